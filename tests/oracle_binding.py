"""ctypes binding of oracle/libchain_oracle.so -- the CPU oracle.  TEST INFRASTRUCTURE: imported only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libchain_oracle.so")


class OParams(C.Structure):
    _fields_ = [("max_dist_x", C.c_int32), ("max_dist_y", C.c_int32), ("bw", C.c_int32), ("max_skip", C.c_int32),
                ("max_iter", C.c_int32), ("gap_scale", C.c_float), ("is_cdna", C.c_int32), ("n_segs", C.c_int32)]


def oparams(p):
    """from an mm2chain Params (or anything with the same field names)"""
    return OParams(p.max_dist_x, p.max_dist_y, p.bw, p.max_skip, p.max_iter, p.gap_scale, p.is_cdna, p.n_segs)


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        vp = C.c_void_p
        lib.mm2o_avg_qspan_scaled.restype = C.c_float
        lib.mm2o_avg_qspan_scaled.argtypes = [C.c_int64, vp]
        lib.mm2o_chain_fpv.restype = None
        lib.mm2o_chain_fpv.argtypes = [C.POINTER(OParams), C.c_int64, vp, C.c_float, vp, vp, vp, vp]
        lib.mm2o_predict.restype = C.c_int64
        lib.mm2o_predict.argtypes = [C.c_int64, vp, C.c_int32, vp, C.POINTER(C.c_int64)]
        lib.mm2o_fill_v.restype = None
        lib.mm2o_fill_v.argtypes = [C.c_int64, vp, vp, vp]
        lib.mm2o_chain_hw_literal.restype = None
        lib.mm2o_chain_hw_literal.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp, vp, vp, vp]
        lib.mm2o_mm_chain_dp.restype = C.c_int32
        lib.mm2o_mm_chain_dp.argtypes = [C.POINTER(OParams), C.c_int32, C.c_int32, C.c_int64, vp, C.POINTER(vp), C.POINTER(vp),
                                         C.POINTER(C.c_int64)]
        lib.mm2o_backtrack.restype = C.c_int32
        lib.mm2o_backtrack.argtypes = [C.c_int64, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(vp),
                                       C.POINTER(C.c_int64)]
        lib.mm2o_bench_batch.restype = C.c_double
        lib.mm2o_bench_batch.argtypes = [C.POINTER(OParams), C.c_int64, vp, vp, vp, vp, C.c_int]
        lib.mm2o_radix_sort_64.argtypes = [vp, C.c_int64]
        lib.mm2o_radix_sort_128x.argtypes = [vp, C.c_int64]
        lib.mm2o_collect_seed_hits.restype = C.c_int64
        lib.mm2o_collect_seed_hits.argtypes = [C.c_int64, vp, vp, C.c_int32, vp]
        lib.mm2o_collect_seed_hits_flags.restype = C.c_int64
        lib.mm2o_collect_seed_hits_flags.argtypes = [C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp]
        lib.mm2o_collect_seed_hits_heap.restype = C.c_int64
        lib.mm2o_collect_seed_hits_heap.argtypes = [C.c_int64, vp, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp]
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_anchor_array(anchors):
    return np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)


def avg_qspan(anchors):
    a = as_anchor_array(anchors)
    return float(load().mm2o_avg_qspan_scaled(a.shape[0], _ptr(a)))


def chain_fpv(par, anchors, avg=None):
    """returns f, p, v for one task (stock CPU semantics)"""
    a = as_anchor_array(anchors)
    n = a.shape[0]
    if avg is None:
        avg = avg_qspan(a) if n else 0.0
    f = np.zeros(n, np.int32); p = np.zeros(n, np.int32); v = np.zeros(n, np.int32); t = np.zeros(n, np.int32)
    op = par if isinstance(par, OParams) else oparams(par)
    load().mm2o_chain_fpv(C.byref(op), n, _ptr(a), avg, _ptr(f), _ptr(p), _ptr(v), _ptr(t))
    return f, p, v


def chain_batch(par, offsets, anchors, n_threads=1):
    """f, p over a CSR batch; returns (f, p, seconds)"""
    a = as_anchor_array(anchors)
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    assert off.size >= 1 and off[0] >= 0 and off[-1] <= a.shape[0] and np.all(np.diff(off) >= 0), "offsets do not fit the anchor array"
    f = np.zeros(a.shape[0], np.int32); p = np.zeros(a.shape[0], np.int32)
    op = par if isinstance(par, OParams) else oparams(par)
    secs = load().mm2o_bench_batch(C.byref(op), off.size - 1, _ptr(off), _ptr(a), _ptr(f), _ptr(p), n_threads)
    return f, p, secs


def predict(anchors, max_dist_x):
    a = as_anchor_array(anchors)
    ns = np.zeros(a.shape[0], np.uint8)
    trip = C.c_int64(0)
    tot = load().mm2o_predict(a.shape[0], _ptr(a), max_dist_x, _ptr(ns), C.byref(trip))
    return ns, int(tot), int(trip.value)


def chain_hw_literal(max_dist_x, max_dist_y, bw, q_span, avg, anchors):
    a = as_anchor_array(anchors)
    n = a.shape[0]
    ns, _, _ = predict(a, max_dist_x)
    f = np.zeros(n, np.int32); p = np.zeros(n, np.int32)
    load().mm2o_chain_hw_literal(n, max_dist_x, max_dist_y, bw, q_span, avg, _ptr(a), _ptr(ns), _ptr(f), _ptr(p))
    return f, p


def mm_chain_dp(par, min_cnt, min_sc, anchors):
    a = as_anchor_array(anchors)
    op = par if isinstance(par, OParams) else oparams(par)
    u = C.c_void_p(0); b = C.c_void_p(0); nb = C.c_int64(0)
    n_u = load().mm2o_mm_chain_dp(C.byref(op), min_cnt, min_sc, a.shape[0], _ptr(a), C.byref(u), C.byref(b), C.byref(nb))
    if n_u == 0:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    u_np = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(n_u,)).copy()
    b_np = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(nb.value, 2)).copy()
    libc.free(u); libc.free(b)
    return u_np, b_np


def backtrack(min_cnt, min_sc, anchors, f, p):
    """the part of mm_chain_dp after the DP (chain.c:106-111 v[] fill, then :348-422) on ANY f[] / p[]: returns u, b"""
    a = as_anchor_array(anchors)
    n = a.shape[0]
    f = np.ascontiguousarray(f, dtype=np.int32); p = np.ascontiguousarray(p, dtype=np.int32)
    assert f.size == n and p.size == n and (n == 0 or (p.max() < n and p.min() >= -1 and np.all(p < np.arange(n))))
    if n == 0:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    v = np.zeros(n, np.int32); t = np.zeros(n, np.int32)
    lib = load()
    lib.mm2o_fill_v(n, _ptr(f), _ptr(p), _ptr(v))
    u = C.c_void_p(0); b = C.c_void_p(0); nb = C.c_int64(0)
    n_u = lib.mm2o_backtrack(n, _ptr(a), min_cnt, min_sc, _ptr(f), _ptr(p), _ptr(v), _ptr(t), C.byref(u), C.byref(b), C.byref(nb))
    if n_u == 0:
        return np.zeros(0, np.uint64), np.zeros((0, 2), np.uint64)
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    u_np = np.ctypeslib.as_array(C.cast(u, C.POINTER(C.c_uint64)), shape=(n_u,)).copy()
    b_np = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(nb.value, 2)).copy()
    libc.free(u); libc.free(b)
    return u_np, b_np


MATCH_DTYPE = np.dtype([("cr_off", "<i8"), ("n", "<u4"), ("q_pos", "<u4"), ("q_span", "<u4"), ("seg_tandem", "<u4")])   # mm2o_match_t / mm2c_match_t


F_NO_DIAG, F_NO_DUAL, F_FOR_ONLY, F_REV_ONLY = 0x001, 0x002, 0x100000, 0x200000      # minimap.h:8-9,28-29


def collect_seed_hits(matches, hits, qlen, flag=0, ref_rank=None, ref_len=None, q_lo=0, q_eq=0, heap=False):
    """collect_seed_hits (map.c:215-247) of one read: matches (MATCH_DTYPE), hit pool (uint64) -> sorted anchors uint64 [n, 2].
    flag / ref_rank / ref_len / q_lo / q_eq: skip_seed (map.c:122-147), names carried by ranks (oracle/chain_oracle.c);
    heap: collect_seed_hits_heap (map.c:149-213, MM_F_HEAP_SORT) instead -- the same anchors, another order among equal x"""
    m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
    h = np.ascontiguousarray(hits, dtype=np.uint64)
    if m.size and int((m["cr_off"] + m["n"]).max()) > h.size:
        raise ValueError("matches reach beyond the hit pool")
    a = np.zeros((max(int(m["n"].sum()), 1), 2), np.uint64)
    rr = np.ascontiguousarray(ref_rank, dtype=np.int32) if ref_rank is not None else None
    rl = np.ascontiguousarray(ref_len, dtype=np.int32) if ref_len is not None else None
    if rr is not None and h.size and int((h >> np.uint64(32)).max()) >= rr.size:
        raise ValueError("a hit names a reference sequence beyond ref_rank")
    fn = load().mm2o_collect_seed_hits_heap if heap else load().mm2o_collect_seed_hits_flags
    n = fn(m.size, _ptr(m), _ptr(h), int(qlen), int(flag), _ptr(rr) if rr is not None else None,
                                            _ptr(rl) if rl is not None else None, int(q_lo), int(q_eq), _ptr(a))
    return a[:n]
