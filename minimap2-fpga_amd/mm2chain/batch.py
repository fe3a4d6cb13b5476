"""Host-side mirror of the reference's dispatch surface for the chaining DP, over the C ABI.

ChainPlan   : a CSR batch of tasks resident in HBM (the throughput path; north-star "many reads in flight").
chain_task  : one task from host memory, the extended run_chaining_on_hw (chain_hardware.h:68 + the five scalars).
run_chaining_on_hw / hardware_init / cleanup : the reference's own entry points, called through their C++ symbols.
mm_chain_dp : the whole reference function (mmpriv.h:65), DP on the GPU, epilogue on the host.
torch is plumbing only: device memory, streams.
"""
import ctypes as C
import os
import numpy as np
import torch

from . import _native as N
from .params import Params

_inited = False


def init(device=None):
    """hardware_init equivalent (main.c:367).  Raises when no HIP device is usable."""
    global _inited
    lib = N.load()
    if device is None:
        # (MM2C_DEVICES in the environment names the devices when the caller names none: mm2c_init(-1), as for a host whose init hook carries no ordinals)
        device = torch.cuda.current_device() if torch.cuda.is_available() and not os.environ.get("MM2C_DEVICES") else -1
    N.check(lib.mm2c_init(int(device)), "mm2c_init")
    _inited = True


def init_devices(ordinals):
    """several devices in one process (the reference's NUM_HW_KERNELS scaffolding, chain_hardware.cpp:9-23): ordinals[0] is the primary
    device; the host-batch entries split big batches across all of them.  An ordinal may repeat."""
    global _inited
    arr = (C.c_int * len(ordinals))(*[int(d) for d in ordinals])
    N.check(N.load().mm2c_init_devices(len(ordinals), arr), "mm2c_init_devices")
    _inited = True


def device_count():
    return N.load().mm2c_device_count()


def split_tasks(offsets, n_parts):
    """mm2c_split_tasks: bounds of n_parts contiguous task ranges with about equal anchor counts (no GPU involved)"""
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    bounds = np.zeros(n_parts + 1, dtype=np.int64)
    N.check(N.load().mm2c_split_tasks(off.size - 1, off.ctypes.data, int(n_parts), bounds.ctypes.data), "mm2c_split_tasks")
    return bounds


def split_model(preset="map-ont"):
    """the HW/SW split constants for this hardware (chain.c:80-81; include/mm2chain_split.h) as a dict K1_HW, K2_HW, C_HW, K_SW, C_SW"""
    v = [C.c_float(0) for _ in range(5)]
    N.check(N.load().mm2c_split_model(preset.encode(), *[C.byref(x) for x in v]), "mm2c_split_model")
    return dict(zip(("K1_HW", "K2_HW", "C_HW", "K_SW", "C_SW"), [x.value for x in v]))


def shutdown():
    """cleanup() equivalent (main.c:430)"""
    global _inited
    if _inited:
        N.load().mm2c_shutdown()
        _inited = False


def tune(key, value):
    N.check(N.load().mm2c_tune(key.encode(), int(value)), "mm2c_tune")


def device_info():
    lib = N.load()
    name = C.create_string_buffer(256)
    cu, mem = C.c_int(0), C.c_size_t(0)
    N.check(lib.mm2c_device_info(name, 256, C.byref(cu), C.byref(mem)), "mm2c_device_info")
    return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}


def last_host_variant():
    """which DP instantiation the last host-buffer entry launched ("chain_dp_tile<...> loop=asm ... compact=1")"""
    buf = C.create_string_buffer(256)
    N.check(N.load().mm2c_last_host_variant(buf, 256), "mm2c_last_host_variant")
    return buf.value.decode()


def device_identity():
    """{"ordinal", "pci_bus_id", "arch"} of the calling thread's library device (mm2c_device_identity)"""
    lib = N.load()
    o = C.c_int(-1); bus = C.create_string_buffer(64); arch = C.create_string_buffer(256)
    N.check(lib.mm2c_device_identity(C.byref(o), bus, 64, arch, 256), "mm2c_device_identity")
    return {"ordinal": o.value, "pci_bus_id": bus.value.decode(), "arch": arch.value.decode()}


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class ChainPlan:
    """mm2c_plan_t: offsets + longest-first order + workspace on the device; run() enqueues the DP."""

    def __init__(self, params: Params, offsets):
        self.lib = N.load()
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
        self.n_tasks = off.size - 1
        self.params = params
        self.handle = self.lib.mm2c_plan_create(C.byref(params), self.n_tasks, _np_ptr(off))
        if not self.handle:
            N.check(-1, "mm2c_plan_create")
        self.total = self.lib.mm2c_plan_total_anchors(self.handle)

    def run(self, anchors: torch.Tensor, f: torch.Tensor, p: torch.Tensor, avg: torch.Tensor = None, stream=None):
        """anchors: int64 [total, 2] on the GPU; f, p: int32 [total] on the GPU.  Asynchronous on `stream`
        (default: torch's current stream, so torch.cuda.Event timing and torch.cuda.synchronize see it)."""
        assert anchors.is_cuda and f.is_cuda and p.is_cuda, "HBM-resident path needs device tensors"
        assert anchors.dtype == torch.int64 and anchors.is_contiguous() and f.dtype == torch.int32 and p.dtype == torch.int32
        assert f.is_contiguous() and p.is_contiguous()
        if avg is not None:
            assert avg.is_cuda and avg.dtype == torch.float32 and avg.is_contiguous()
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        # the extents of the tensors go with the pointers: the library refuses buffers shorter than the plan (cf. chain_hardware.cpp:34-37)
        N.check(self.lib.mm2c_plan_run_device_n(self.handle, anchors.data_ptr(), anchors.numel() // 2, avg.data_ptr() if avg is not None else None,
                                                avg.numel() if avg is not None else 0, f.data_ptr(), f.numel(), p.data_ptr(), p.numel(), st),
                "mm2c_plan_run_device_n")

    def set_device_offsets(self, offsets: torch.Tensor = None):
        """task sizes that only the device knows (SeedPlan.run_skip): the kernels take the CSR offsets from this int64 device tensor
        [n_tasks + 1]; None goes back to the plan's own.  The tensor must stay alive while the plan uses it."""
        if offsets is not None:
            assert offsets.is_cuda and offsets.dtype == torch.int64 and offsets.numel() == self.n_tasks + 1 and offsets.is_contiguous()
        self._dev_off = offsets
        N.check(self.lib.mm2c_plan_set_device_offsets(self.handle, offsets.data_ptr() if offsets is not None else None), "mm2c_plan_set_device_offsets")

    def predict(self, anchors: torch.Tensor, stream=None):
        """chain.c:53-78 on the GPU: returns (num_subparts uint8 [total], total_subparts int64 [n_tasks],
        total_trip_count int64 [n_tasks]) as device tensors"""
        assert anchors.is_cuda and anchors.dtype == torch.int64 and anchors.numel() == 2 * self.total
        ns = torch.empty(self.total, dtype=torch.uint8, device=anchors.device)
        ts = torch.empty(self.n_tasks, dtype=torch.int64, device=anchors.device)
        tt = torch.empty(self.n_tasks, dtype=torch.int64, device=anchors.device)
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        N.check(self.lib.mm2c_plan_predict_device(self.handle, anchors.data_ptr(), ns.data_ptr(), ts.data_ptr(), tt.data_ptr(), st),
                "mm2c_plan_predict_device")
        return ns, ts, tt

    def chains(self, anchors: torch.Tensor, f: torch.Tensor, p: torch.Tensor, min_cnt: int, min_sc: int, stream=None):
        """the epilogue of mm_chain_dp (chain.c:106-111,348-422) on the GPU, after run() on the same stream.  Returns device
        tensors (u_off int64 [n_tasks+1], u int64 [total], b_off int64 [n_tasks+1], b int64 [total, 2]); only the first
        u_off[-1] / b_off[-1] entries of u / b are defined."""
        assert anchors.is_cuda and anchors.dtype == torch.int64 and f.dtype == torch.int32 and p.dtype == torch.int32
        dev = anchors.device
        u_off = torch.empty(self.n_tasks + 1, dtype=torch.int64, device=dev)
        b_off = torch.empty(self.n_tasks + 1, dtype=torch.int64, device=dev)
        u = torch.empty(max(self.total, 1), dtype=torch.int64, device=dev)
        b = torch.empty((max(self.total, 1), 2), dtype=torch.int64, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        N.check(self.lib.mm2c_plan_chains_device_n(self.handle, anchors.data_ptr(), anchors.numel() // 2, f.data_ptr(), f.numel(), p.data_ptr(), p.numel(),
                                                   min_cnt, min_sc, u_off.data_ptr(), u_off.numel(), u.data_ptr(), u.numel(), b_off.data_ptr(),
                                                   b_off.numel(), b.data_ptr(), b.numel() // 2, st), "mm2c_plan_chains_device_n")
        return u_off, u, b_off, b

    def last_epilogue_ms(self):
        ms = C.c_float(0)
        N.check(self.lib.mm2c_plan_last_epilogue_ms(self.handle, C.byref(ms)), "mm2c_plan_last_epilogue_ms")
        return ms.value

    def last_kernel_ms(self):
        ms = C.c_float(0)
        N.check(self.lib.mm2c_plan_last_kernel_ms(self.handle, C.byref(ms)), "mm2c_plan_last_kernel_ms")
        return ms.value

    def last_route(self):
        """(pieces, pieces run with one wave each, pieces run with sixteen waves each) of the last run -- mm2c_plan_last_route"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        N.check(self.lib.mm2c_plan_last_route(self.handle, C.byref(a), C.byref(b), C.byref(c)), "mm2c_plan_last_route")
        return int(a.value), int(b.value), int(c.value)

    def last_variant(self):
        """text naming the kernel instantiation the last run launched (mm2c_plan_last_variant)"""
        buf = C.create_string_buffer(192)
        N.check(self.lib.mm2c_plan_last_variant(self.handle, buf, len(buf)), "mm2c_plan_last_variant")
        return buf.value.decode()

    def last_prepass_ms(self):
        ms = C.c_float(0)
        N.check(self.lib.mm2c_plan_last_prepass_ms(self.handle, C.byref(ms)), "mm2c_plan_last_prepass_ms")
        return ms.value

    def close(self):
        if self.handle:
            self.lib.mm2c_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedArray:
    """numpy view of page-locked host memory from mm2c_pinned_alloc (freed when the object dies)"""

    def __init__(self, shape, dtype):
        self.lib = N.load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.lib.mm2c_pinned_alloc(self.nbytes)
        if not self.ptr:
            N.check(-1, "mm2c_pinned_alloc")
        buf = (C.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        try:
            if self.ptr:
                self.array = None
                self.lib.mm2c_pinned_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def chain_batch_host_into(params: Params, offsets, anchors, f, p, avg=None):
    """as chain_batch_host, but into caller-provided int32 arrays (e.g. PinnedArray.array); anchors may be pinned too"""
    lib = N.load()
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    a = anchors.view(np.uint64).reshape(-1, 2)
    assert a.flags.c_contiguous and f.flags.c_contiguous and p.flags.c_contiguous and f.dtype == np.int32 and p.dtype == np.int32
    if off.size and (off[0] < 0 or off[-1] > a.shape[0] or off[-1] > f.size or off[-1] > p.size):
        raise ValueError("offsets do not fit the arrays")
    avg_p = None
    if avg is not None:
        avg = np.ascontiguousarray(avg, dtype=np.float32)
        avg_p = _np_ptr(avg)
    N.check(lib.mm2c_chain_batch_host(C.byref(params), off.size - 1, _np_ptr(off), _np_ptr(a), avg_p, _np_ptr(f), _np_ptr(p)),
            "mm2c_chain_batch_host")


def chain_batch_host(params: Params, offsets, anchors, avg=None):
    """Whole batch from host numpy arrays (uint64 [total,2]); returns (f, p) int32 arrays.  PCIe included."""
    lib = N.load()
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    total = int(off[-1] - off[0]) if off.size else 0
    if off.size and (off[0] < 0 or off[-1] > a.shape[0]):
        raise ValueError("offsets do not fit the anchor array")
    f = np.empty(int(off[-1]) if off.size else 0, dtype=np.int32)
    p = np.empty_like(f)
    avg_p = None
    if avg is not None:
        avg = np.ascontiguousarray(avg, dtype=np.float32)
        avg_p = _np_ptr(avg)
    N.check(lib.mm2c_chain_batch_host(C.byref(params), off.size - 1, _np_ptr(off), _np_ptr(a), avg_p, _np_ptr(f), _np_ptr(p)),
            "mm2c_chain_batch_host")
    del total
    return f, p


def chain_task(params: Params, anchors, avg_qspan_scaled, tid=0):
    """One task, synchronous, stock-CPU (V1) semantics: the extended run_chaining_on_hw."""
    lib = N.load()
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    n = a.shape[0]
    f = np.empty(n, dtype=np.int32)
    p = np.empty(n, dtype=np.int32)
    N.check(lib.mm2c_chain_task_host(C.byref(params), n, _np_ptr(a), float(avg_qspan_scaled), _np_ptr(f), _np_ptr(p), tid),
            "mm2c_chain_task_host")
    return f, p


def chain_task_pred(params: Params, anchors, avg_qspan_scaled, tid, hw_time_pred, sw_time_pred):
    """mm2c_chain_task_host_pred: the same call under the reference's busy protocol (chain_hardware.cpp:54-75).  Returns (ret, f, p); ret 1 = declined, f / p untouched."""
    lib = N.load()
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    n = a.shape[0]
    f = np.full(n, -77, dtype=np.int32)
    p = np.full(n, -77, dtype=np.int32)
    rc = lib.mm2c_chain_task_host_pred(C.byref(params), n, _np_ptr(a), float(avg_qspan_scaled), _np_ptr(f), _np_ptr(p), tid, float(hw_time_pred), float(sw_time_pred))
    if rc != 1:
        N.check(rc, "mm2c_chain_task_host_pred")
    return rc, f, p


def slot_stats(slot):
    """mm2c_get_slot_stats as a dict: what the call combiner of device slot `slot` has served since init"""
    st = N.SlotStats()
    N.check(N.load().mm2c_get_slot_stats(int(slot), C.byref(st)), "mm2c_get_slot_stats")
    return {k: int(getattr(st, k)) for k, _ in st._fields_ if k != "reserved"}


def _split_chains(n_tasks, u_off, u, b_off, b):
    return [(u[u_off[k]:u_off[k + 1]].copy(), b[b_off[k]:b_off[k + 1]].copy()) for k in range(n_tasks)]


def mm_chain_dp_batch(params: Params, min_cnt, min_sc, offsets, anchors, epilogue_threads=0):
    """mm2c_mm_chain_dp_batch_host: per-task list [(u uint64 [n_u], b uint64 [n_b, 2]), ...]; epilogue_threads == 0 runs the
    epilogue on the GPU, > 0 on that many host threads"""
    lib = N.load()
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    n_tasks, total = off.size - 1, int(off[-1] - off[0])
    if off[-1] > a.shape[0] or off[0] < 0:
        raise ValueError("offsets reach beyond the anchor array")
    u_off = np.zeros(n_tasks + 1, np.int64); b_off = np.zeros(n_tasks + 1, np.int64)
    u = np.zeros(max(total, 1), np.uint64); b = np.zeros((max(total, 1), 2), np.uint64)
    N.check(lib.mm2c_mm_chain_dp_batch_host(C.byref(params), min_cnt, min_sc, n_tasks, _np_ptr(off), _np_ptr(a), epilogue_threads,
                                            _np_ptr(u_off), _np_ptr(u), _np_ptr(b_off), _np_ptr(b)), "mm2c_mm_chain_dp_batch_host")
    return _split_chains(n_tasks, u_off, u, b_off, b)


def chain_epilogue_host(min_cnt, min_sc, offsets, anchors, f, p, n_threads=4):
    """mm2c_chain_epilogue_host: the epilogue on host threads from f[] / p[] (no GPU involved)"""
    lib = N.load()
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    f = np.ascontiguousarray(f, dtype=np.int32); p = np.ascontiguousarray(p, dtype=np.int32)
    n_tasks, total = off.size - 1, int(off[-1] - off[0])
    if off[-1] > a.shape[0] or off[0] < 0 or f.size < off[-1] or p.size < off[-1]:
        raise ValueError("offsets reach beyond the arrays")
    u_off = np.zeros(n_tasks + 1, np.int64); b_off = np.zeros(n_tasks + 1, np.int64)
    u = np.zeros(max(total, 1), np.uint64); b = np.zeros((max(total, 1), 2), np.uint64)
    N.check(lib.mm2c_chain_epilogue_host(min_cnt, min_sc, n_tasks, _np_ptr(off), _np_ptr(a), _np_ptr(f), _np_ptr(p), n_threads,
                                         _np_ptr(u_off), _np_ptr(u), _np_ptr(b_off), _np_ptr(b)), "mm2c_chain_epilogue_host")
    return _split_chains(n_tasks, u_off, u, b_off, b)


MATCH_DTYPE = np.dtype([("cr_off", "<i8"), ("n", "<u4"), ("q_pos", "<u4"), ("q_span", "<u4"), ("seg_tandem", "<u4")])   # mm2c_match_t


class SeedPlan:
    """mm2c_seedplan_t: seed hits -> sorted anchors on the GPU (collect_seed_hits, map.c:215-247) for a batch of reads"""

    def __init__(self, match_off, anchor_off):
        self.lib = N.load()
        mo = np.ascontiguousarray(np.asarray(match_off, dtype=np.int64)); ao = np.ascontiguousarray(np.asarray(anchor_off, dtype=np.int64))
        assert mo.size == ao.size
        self.n_reads, self.total = mo.size - 1, int(ao[-1] - ao[0])
        self.handle = self.lib.mm2c_seedplan_create(self.n_reads, _np_ptr(mo), _np_ptr(ao))
        if not self.handle:
            N.check(-1, "mm2c_seedplan_create")

    def run(self, matches: torch.Tensor, hits: torch.Tensor, qlen: torch.Tensor, anchors: torch.Tensor = None, stream=None):
        """matches: uint8 view of mm2c_match_t records on the GPU; hits: int64 pool; qlen: int32 [n_reads]; returns anchors int64 [total, 2]"""
        assert matches.is_cuda and hits.is_cuda and qlen.is_cuda and qlen.dtype == torch.int32 and qlen.numel() == self.n_reads
        if anchors is None:
            anchors = torch.empty((max(self.total, 1), 2), dtype=torch.int64, device=hits.device)
        assert anchors.is_cuda and anchors.dtype == torch.int64 and anchors.numel() >= 2 * self.total
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        N.check(self.lib.mm2c_seedplan_run_device_n(self.handle, matches.data_ptr(), matches.numel() * matches.element_size() // 24, hits.data_ptr(),
                                                    hits.numel(), qlen.data_ptr(), qlen.numel(), anchors.data_ptr(), anchors.numel() // 2, st),
                "mm2c_seedplan_run_device_n")
        return anchors

    def run_skip(self, matches, hits, qlen, flag, ref_rank, ref_len, q_lo, q_eq, anchors=None, stream=None):
        """collect_seed_hits with skip_seed (map.c:122-147; -x ava-ont: flag = NO_DIAG | NO_DUAL): the reads keep fewer anchors than they have
        hits.  ref_rank / ref_len: int32 per reference sequence, q_lo / q_eq: int32 per read (device tensors; names as ranks, mm2chain.h).
        Returns (anchors int64 [capacity, 2] packed, offsets int64 [n_reads + 1] on the device)."""
        assert matches.is_cuda and hits.is_cuda and qlen.is_cuda and qlen.dtype == torch.int32 and qlen.numel() == self.n_reads
        for t in (ref_rank, ref_len, q_lo, q_eq):
            assert t is None or (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous())
        if anchors is None:
            anchors = torch.empty((max(self.total, 1), 2), dtype=torch.int64, device=hits.device)
        off = torch.empty(self.n_reads + 1, dtype=torch.int64, device=hits.device)
        sk = N.SeedSkip(int(flag), ref_rank.data_ptr() if ref_rank is not None else None, ref_len.data_ptr() if ref_len is not None else None,
                        q_lo.data_ptr() if q_lo is not None else None, q_eq.data_ptr() if q_eq is not None else None)
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        N.check(self.lib.mm2c_seedplan_run_device_skip(self.handle, matches.data_ptr(), matches.numel() * matches.element_size() // 24, hits.data_ptr(),
                                                       hits.numel(), qlen.data_ptr(), qlen.numel(), C.byref(sk), anchors.data_ptr(),
                                                       anchors.numel() // 2, off.data_ptr(), st), "mm2c_seedplan_run_device_skip")
        return anchors, off

    def set_heap_sort(self, on=True):
        """MM_F_HEAP_SORT (--heap-sort, -x sr): later runs leave the order of collect_seed_hits_heap (map.c:149-213) among anchors with equal x"""
        N.check(self.lib.mm2c_seedplan_set_heap_sort(self.handle, 1 if on else 0), "mm2c_seedplan_set_heap_sort")

    def check(self):
        """waits for the run; raises if a read's hit counts did not add up to its anchor range; returns the number of reads with equal x"""
        n = C.c_int64(0)
        N.check(self.lib.mm2c_seedplan_check(self.handle, C.byref(n)), "mm2c_seedplan_check")
        return n.value

    def last_ms(self):
        ms = C.c_float(0)
        N.check(self.lib.mm2c_seedplan_last_ms(self.handle, C.byref(ms)), "mm2c_seedplan_last_ms")
        return ms.value

    def close(self):
        if self.handle:
            self.lib.mm2c_seedplan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def seed_hits_batch(match_off, matches, hits, qlen):
    """mm2c_seed_hits_batch_host: returns (anchor_off int64 [n_reads+1], anchors uint64 [total, 2])"""
    lib = N.load()
    mo = np.ascontiguousarray(np.asarray(match_off, dtype=np.int64))
    m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
    h = np.ascontiguousarray(hits, dtype=np.uint64)
    q = np.ascontiguousarray(qlen, dtype=np.int32)
    n_reads = mo.size - 1
    if q.size != n_reads or mo[-1] > m.size or mo[0] < 0:
        raise ValueError("offsets do not fit the arrays")
    ao = np.zeros(n_reads + 1, np.int64)
    a = np.zeros((max(int(m["n"][mo[0]:mo[-1]].sum()), 1), 2), np.uint64)
    N.check(lib.mm2c_seed_hits_batch_host(n_reads, _np_ptr(mo), _np_ptr(m), _np_ptr(h), h.size, _np_ptr(q), _np_ptr(ao), _np_ptr(a)),
            "mm2c_seed_hits_batch_host")
    return ao, a[:ao[-1]]


def seed_chain_batch(params: Params, min_cnt, min_sc, match_off, matches, hits, qlen):
    """mm2c_seed_chain_batch_host: matches in, per-read chains out [(u, b), ...] (anchors never leave the GPU)"""
    lib = N.load()
    mo = np.ascontiguousarray(np.asarray(match_off, dtype=np.int64))
    m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
    h = np.ascontiguousarray(hits, dtype=np.uint64)
    q = np.ascontiguousarray(qlen, dtype=np.int32)
    n_reads = mo.size - 1
    if q.size != n_reads or mo[-1] > m.size or mo[0] < 0:
        raise ValueError("offsets do not fit the arrays")
    total = max(int(m["n"][mo[0]:mo[-1]].sum()), 1)
    ao = np.zeros(n_reads + 1, np.int64); u_off = np.zeros(n_reads + 1, np.int64); b_off = np.zeros(n_reads + 1, np.int64)
    u = np.zeros(total, np.uint64); b = np.zeros((total, 2), np.uint64)
    N.check(lib.mm2c_seed_chain_batch_host(C.byref(params), min_cnt, min_sc, n_reads, _np_ptr(mo), _np_ptr(m), _np_ptr(h), h.size, _np_ptr(q),
                                           _np_ptr(ao), _np_ptr(u_off), _np_ptr(u), _np_ptr(b_off), _np_ptr(b)), "mm2c_seed_chain_batch_host")
    return _split_chains(n_reads, u_off, u, b_off, b)


class HitPool:
    """mm2c_hitpool_t: the index's position arrays resident in HBM (uploaded once per index)"""

    def __init__(self, hits):
        self.lib = N.load()
        h = np.ascontiguousarray(hits, dtype=np.uint64)
        self.handle = self.lib.mm2c_hitpool_create(_np_ptr(h), h.size)
        if not self.handle:
            N.check(-4, "mm2c_hitpool_create")
        self.size = h.size

    def close(self):
        if self.handle:
            self.lib.mm2c_hitpool_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def seed_chain_batch_pool(params: Params, min_cnt, min_sc, match_off, matches, pool: HitPool, qlen):
    """mm2c_seed_chain_batch_pool: as seed_chain_batch with the hits taken from a resident pool (cr_off points into it)"""
    lib = N.load()
    mo = np.ascontiguousarray(np.asarray(match_off, dtype=np.int64))
    m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
    q = np.ascontiguousarray(qlen, dtype=np.int32)
    n_reads = mo.size - 1
    if q.size != n_reads or mo[-1] > m.size or mo[0] < 0:
        raise ValueError("offsets do not fit the arrays")
    total = max(int(m["n"][mo[0]:mo[-1]].sum()), 1)
    ao = np.zeros(n_reads + 1, np.int64); u_off = np.zeros(n_reads + 1, np.int64); b_off = np.zeros(n_reads + 1, np.int64)
    u = np.zeros(total, np.uint64); b = np.zeros((total, 2), np.uint64)
    N.check(lib.mm2c_seed_chain_batch_pool(C.byref(params), min_cnt, min_sc, n_reads, _np_ptr(mo), _np_ptr(m), pool.handle, _np_ptr(q),
                                           _np_ptr(ao), _np_ptr(u_off), _np_ptr(u), _np_ptr(b_off), _np_ptr(b)), "mm2c_seed_chain_batch_pool")
    return _split_chains(n_reads, u_off, u, b_off, b)


def stage_stats(reset=False):
    """mm2c_get_stage_stats as a dict (ns and counts); reset=True clears the counters afterwards"""
    lib = N.load()
    st = N.StageStats()
    lib.mm2c_get_stage_stats(C.byref(st))
    if reset:
        lib.mm2c_reset_stage_stats()
    return {k: int(getattr(st, k)) for k, _ in st._fields_}


def hardware_init(buf_size=0, binary_name=b""):
    """the reference symbol bool hardware_init(long, char*) (chain_hardware.h:70)"""
    return bool(getattr(N.load(), "_Z13hardware_initlPc")(buf_size, binary_name))


def cleanup():
    """the reference symbol void cleanup() (chain_hardware.h:71)"""
    getattr(N.load(), "_Z7cleanupv")()


def run_chaining_on_hw(n, max_dist_x, max_dist_y, bw, q_span, avg_qspan, anchors, num_subparts=None, total_subparts=0, tid=0,
                       hw_time_pred=0.0, sw_time_pred=0.0):
    """the reference symbol (chain_hardware.h:68), same argument order; returns (ret, f, p)"""
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    f = np.empty(n, dtype=np.int32)
    p = np.empty(n, dtype=np.int32)
    ns = _np_ptr(np.ascontiguousarray(num_subparts, dtype=np.uint8)) if num_subparts is not None else None
    ret = getattr(N.load(), "_Z18run_chaining_on_hwliiiifP7mm128_tPiS1_Phliff")(
        n, max_dist_x, max_dist_y, bw, q_span, avg_qspan, _np_ptr(a), _np_ptr(f), _np_ptr(p), ns, total_subparts, tid,
        hw_time_pred, sw_time_pred)
    return ret, f, p


def mm_chain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale, is_cdna, n_segs, anchors, tid=0):
    """mm_chain_dp (mmpriv.h:65) through the library's host mirror.  km = NULL (malloc arena, as kalloc.c does).
    Returns (u uint64 [n_u], b uint64 [sum cnt, 2])."""
    lib = N.load()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    n = a.shape[0]
    a_own = libc.malloc(max(n, 1) * 16)       # mm_chain_dp frees its input (chain.c:39,421)
    C.memmove(a_own, a.ctypes.data, n * 16)
    n_u = C.c_int(0)
    u = C.c_void_p(0)
    b = lib.mm_chain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale, is_cdna, n_segs,
                        n, a_own if n else None, C.byref(n_u), C.byref(u), None, tid)
    if n == 0:
        libc.free(a_own)
    if not b or n_u.value == 0:
        if b:
            libc.free(b)
        if u.value:
            libc.free(u)
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    u_np = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(n_u.value,)).copy()
    nb = int((u_np & 0xFFFFFFFF).sum())
    b_np = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(nb, 2)).copy()
    libc.free(b)
    libc.free(u)
    return u_np, b_np
