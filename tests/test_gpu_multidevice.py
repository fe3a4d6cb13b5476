"""In-process multi-device path of the library (mm2c_init_devices; the reference scaffolds per-kernel queues / buffers / locks,
chain_hardware.cpp:9-23).  A GPU box of the test pool has one MI355X, so the same ordinal is listed twice (and three times): two device
contexts with their own streams and arenas, host batches split into contiguous task ranges that run side by side on worker threads and
are closed up afterwards -- everything the 8-GPU form does except that the contexts share a device.  Results must equal the
single-context results (which the other GPU tests pin to the oracle)."""
import numpy as np
import pytest
import torch

import oracle_binding as ob
from helpers import oracle_batch, assert_same

pytestmark = pytest.mark.gpu


def _stream(profile, n_reads, n_per, seed):
    from mm2chain import synth
    off, a = synth.make_stream(profile, n_reads, n_per, seed=seed)
    return off.numpy(), a.numpy().view(np.uint64)


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_host_batches_split_across_device_contexts(n_ctx):
    import mm2chain
    from mm2chain import params, synth
    assert torch.cuda.is_available()
    P = params.map_ont()
    off, a = _stream("mixed", 300, (200, 6000), seed=31 + n_ctx)
    off = off + 5                                              # offsets need not start at 0
    a_all = np.concatenate([np.zeros((5, 2), np.uint64), a])
    f_ref, p_ref = oracle_batch(P, off - 5, a)
    mm2chain.shutdown()
    mm2chain.init_devices([0] * n_ctx)
    try:
        assert mm2chain.device_count() == n_ctx
        mm2chain.tune("multi_min_anchors", 1000)
        # f / p through the host-buffer path
        f, p = mm2chain.chain_batch_host(P, off, a_all)
        assert_same(f[5:], p[5:], f_ref, p_ref, off - 5, "split chain_batch_host")
        # whole mm_chain_dp: DP + epilogue on the GPU, chains closed up across the ranges
        res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a_all, epilogue_threads=0)
        for k in (0, 1, 57, 150, 299):
            u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, a_all[off[k]:off[k + 1]])
            assert np.array_equal(res[k][0], u_ref) and np.array_equal(res[k][1], b_ref), k
        n_chains = sum(r[0].size for r in res)
        # matches in, chains out
        ms, hs, mo, ho, ql = [], [], [0], [0], []
        for k in range(40):
            m, h = synth.matches_from_anchors(a_all[off[k]:off[k + 1]], 1 << 20)
            m = m.copy(); m["cr_off"] += ho[-1]
            ms.append(m); hs.append(h); mo.append(mo[-1] + m.size); ho.append(ho[-1] + h.size); ql.append(1 << 20)
        out = mm2chain.seed_chain_batch(P, 3, 40, np.array(mo, np.int64), np.concatenate(ms), np.concatenate(hs), np.array(ql, np.int32))
    finally:
        mm2chain.shutdown()
    # the same three calls with one context
    mm2chain.init(0)
    try:
        res1 = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a_all, epilogue_threads=0)
        out1 = mm2chain.seed_chain_batch(P, 3, 40, np.array(mo, np.int64), np.concatenate(ms), np.concatenate(hs), np.array(ql, np.int32))
    finally:
        mm2chain.shutdown()
    assert n_chains == sum(r[0].size for r in res1)
    for k in range(len(res)):
        assert np.array_equal(res[k][0], res1[k][0]) and np.array_equal(res[k][1], res1[k][1]), k
    assert len(out) == len(out1) == 40
    for k in range(40):
        assert np.array_equal(out[k][0], out1[k][0]) and np.array_equal(out[k][1], out1[k][1]), k
        assert np.array_equal(out[k][0], res1[k][0]), k          # seeds -> chains == anchors -> chains (scores and counts; the order among equal x is the sort's)


def test_small_ranges_of_a_split_batch_stay_on_their_own_device_context():
    """ranges of at most combine_max_anchors (2^17) anchors used to go through the call combiner, whose stream and arenas belong to the primary
    device, from worker threads bound to other devices; a worker now always runs on the context of its own device slot.  Small batch, three
    contexts, per-read calls (which do use the combiner) before and after, passes counted."""
    import mm2chain
    from mm2chain import params, _native as N
    P = params.map_ont()
    off, a = _stream("mixed", 60, (200, 3000), seed=5)
    assert int(off[-1]) < 3 * (1 << 17)
    f_ref, p_ref = oracle_batch(P, off, a)
    mm2chain.shutdown()
    mm2chain.init_devices([0, 0, 0])
    try:
        mm2chain.tune("multi_min_anchors", 1000)
        k = 7
        f1, p1 = mm2chain.chain_task(P, a[off[k]:off[k + 1]], ob.avg_qspan(a[off[k]:off[k + 1]]))       # creates the combiner's context
        assert_same(f1, p1, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, "per-read call before the split batch")
        st0 = N.Stats(); N.load().mm2c_get_stats(st0)
        f, p = mm2chain.chain_batch_host(P, off, a)
        st1 = N.Stats(); N.load().mm2c_get_stats(st1)
        assert_same(f, p, f_ref, p_ref, off, "split batch with ranges below combine_max_anchors")
        assert st1.passes - st0.passes == 3                      # one pass per device context, none through the combiner
        f1, p1 = mm2chain.chain_task(P, a[off[k]:off[k + 1]], ob.avg_qspan(a[off[k]:off[k + 1]]))
        assert_same(f1, p1, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, "per-read call after the split batch")
    finally:
        mm2chain.shutdown()
        mm2chain.init(0)


def test_devices_named_in_the_environment_for_hosts_whose_init_hook_carries_no_ordinals(monkeypatch):
    """hardware_init(long, char *) (chain_hardware.h:69) -> mm2c_init(-1): MM2C_DEVICES names the devices"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream("mixed", 120, (200, 3000), seed=77)
    f_ref, p_ref = oracle_batch(P, off, a)
    mm2chain.shutdown()
    try:
        monkeypatch.setenv("MM2C_DEVICES", "0,0")
        mm2chain.init(-1)
        assert mm2chain.device_count() == 2
        mm2chain.tune("multi_min_anchors", 1000)
        f, p = mm2chain.chain_batch_host(P, off, a)
        assert_same(f, p, f_ref, p_ref, off, "MM2C_DEVICES=0,0")
        mm2chain.shutdown()
        monkeypatch.setenv("MM2C_DEVICES", "all")
        mm2chain.init(-1)
        assert mm2chain.device_count() == torch.cuda.device_count()
        mm2chain.shutdown()
        monkeypatch.setenv("MM2C_DEVICES", "0,99")
        with pytest.raises(Exception):
            mm2chain.init(-1)
    finally:
        monkeypatch.delenv("MM2C_DEVICES", raising=False)
        mm2chain.shutdown()
        mm2chain.init(0)


def test_bench_gpus_2_starts_two_ranks_on_the_gpu():
    """`python bench.py --gpus 2` (no launcher): two rank processes, here sharing the box's one MI355X (MM2C_BENCH_ONE_DEVICE) and talking gloo
    (RCCL needs one GPU per rank); each runs the real timed loop on its own batch, the line reports n_gpus 2 and both ranks' checks"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MM2C_BENCH_BACKEND="gloo", MM2C_BENCH_ONE_DEVICE="1")
    for extra in ([], ["--strong", "--ragged"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "512", "--distinct", "256",
                            "--anchors-per-read", "2000", "--cpu-seconds", "0"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and len(out["per_rank_ms_per_step"]) == 2
        assert out["verified_vs_oracle"] is True and out["value"] > 0
        assert out["scaling"] == ("strong" if extra else "weak")


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_per_read_calls_are_served_by_every_device_context(n_ctx):
    """The reference's dispatch surface -- one blocking call per read from many host threads (map.c:561 -> chain.c:103 -> run_chaining_on_hw) -- with several
    devices configured: every device slot has its own call combiner (lanes, streams, arenas on that device) and a call goes to the least loaded slot
    (chain_hardware.cpp:9-23,58-72: a queue, a lock, a buffer set per kernel, one picked per call).  The box has one GPU, listed n_ctx times.  Twelve threads,
    both entries (the reference symbol = V2, the extended entry = V1); every slot must have served passes, and every result must equal the oracle's."""
    import threading
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream("mixed", 96, (150, 5000), seed=91 + n_ctx)
    f_ref, p_ref = oracle_batch(P, off, a)
    Pv2 = params.make_params(max_skip=2**31 - 1, max_iter=1024)
    f_v2, p_v2 = oracle_batch(Pv2, off, a)
    mm2chain.shutdown()
    mm2chain.init_devices([0] * n_ctx)
    errs = []
    try:
        assert mm2chain.device_count() == n_ctx
        mm2chain.tune("multi_min_anchors", 1 << 20)                 # (the library's default; the tests above lower it, and knobs outlive a shutdown)

        def worker(tid):
            try:
                for rep in range(3):
                    for k in range(tid, 96, 12):
                        t = a[off[k]:off[k + 1]]
                        avg = ob.avg_qspan(t)
                        f, p = mm2chain.chain_task(P, t, avg, tid=tid)
                        assert_same(f, p, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, f"V1 read {k} thread {tid}")
                        ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 15, avg, t, None, 0, tid=tid)
                        assert ret == 0
                        assert_same(f, p, f_v2[off[k]:off[k + 1]], p_v2[off[k]:off[k + 1]], None, f"V2 read {k} thread {tid}")
            except Exception as e:                                  # noqa: BLE001 -- reported below, on the main thread
                errs.append(repr(e))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs[:3]
        st = [mm2chain.slot_stats(s) for s in range(n_ctx)]
        assert all(s["device"] == 0 for s in st)
        assert sum(s["calls"] for s in st) == 96 * 3 * 2, st
        assert sum(s["anchors"] for s in st) == int(off[-1]) * 3 * 2, st
        assert all(s["passes"] > 0 and s["calls"] > 0 for s in st), f"a device slot served nothing: {st}"
        with pytest.raises(Exception):
            mm2chain.slot_stats(n_ctx)
    finally:
        mm2chain.shutdown()
        mm2chain.init(0)


def test_busy_protocol_of_the_reference_symbol():
    """chain_hardware.cpp:54-75 (PROCESS_ON_SW_IF_HW_BUSY): run_chaining_on_hw returns 1 = declined when waiting for the device would take longer than sw_time_pred; the
    caller's own loop (chain.c:106,112-164) then runs.  Round 6: OFF by default (a chain.o built without that flag ignores the 1 and would chain from uninitialised f / p,
    chain.c:105,163-169) -- a host opts in with MM2C_DECLINE_WHEN_BUSY / mm2c_tune("decline_when_busy", 1 | 2).  Rule 2 (round 5's: booked predictions): a prediction pair
    the idle device cannot meet is declined with f / p untouched, one it can meet is computed (== the oracle), predictions that are not positive never decline.  Rule 1
    (measured: calls inside the slot and the service time of its passes): a caller whose own loop would take a microsecond is turned away, one whose loop takes seconds
    is served."""
    import mm2chain
    from mm2chain import params
    off, a = _stream("mixed", 2, 1500, seed=12)
    t = a[off[0]:off[1]]
    avg = ob.avg_qspan(t)
    Pv2 = params.make_params(max_skip=2**31 - 1, max_iter=1024)
    f_ref, p_ref, _ = ob.chain_fpv(Pv2, t, avg)
    P = params.map_ont()
    f1, p1, _ = ob.chain_fpv(P, t, avg)
    # default: never declined, whatever the predictions say
    d0 = mm2chain.slot_stats(0)["declined"]
    ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 15, avg, t, None, 0, tid=3, hw_time_pred=2.0, sw_time_pred=0.5)
    assert ret == 0 and mm2chain.slot_stats(0)["declined"] == d0
    assert_same(f, p, f_ref, p_ref, None, "default: the protocol is off")
    try:
        mm2chain.tune("decline_when_busy", 2)
        ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 15, avg, t, None, 0, tid=3, hw_time_pred=2.0, sw_time_pred=0.5)
        assert ret == 1 and mm2chain.slot_stats(0)["declined"] == d0 + 1
        ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 15, avg, t, None, 0, tid=3, hw_time_pred=0.4, sw_time_pred=0.5)
        assert ret == 0
        assert_same(f, p, f_ref, p_ref, None, "accepted call")
        ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 15, avg, t, None, 0, tid=3, hw_time_pred=0.0, sw_time_pred=-1.0)
        assert ret == 0
        assert_same(f, p, f_ref, p_ref, None, "no model: never declined")
        rc, f, p = mm2chain.chain_task_pred(P, t, avg, 0, 9.0, 1.0)
        assert rc == 1 and np.all(f == -77) and np.all(p == -77)          # declined: nothing written
        # rule 1: the slot has served passes by now, so it knows what one takes (tens of microseconds to a millisecond)
        mm2chain.tune("decline_when_busy", 1)
        for _ in range(12):                                               # (the estimate leaves out a slot's first eight passes: they pay for code loading and arena growth)
            mm2chain.chain_task(P, t, avg)
        rc, f, p = mm2chain.chain_task_pred(P, t, avg, 0, 0.05, 1e-6)     # the caller's loop: a nanosecond -- waiting for a pass cannot beat it
        assert rc == 1 and np.all(f == -77) and np.all(p == -77)
        rc, f, p = mm2chain.chain_task_pred(P, t, avg, 0, 0.05, 5000.0)   # the caller's loop: seconds
        assert rc == 0
        assert_same(f, p, f1, p1, None, "rule 1: served")
        rc, f, p = mm2chain.chain_task_pred(P, t, avg, 0, 0.0, 0.0)       # no model
        assert rc == 0
        assert_same(f, p, f1, p1, None, "rule 1, no model: never declined")
    finally:
        mm2chain.tune("decline_when_busy", 0)
    rc, f, p = mm2chain.chain_task_pred(P, t, avg, 0, 9.0, 1.0)
    assert rc == 0
    assert_same(f, p, f1, p1, None, "protocol off")


@pytest.mark.parametrize("direct", [2, 1, 0])
def test_small_host_passes_staged_by_kernels_or_by_copy_commands(direct):
    """Round 5: a staged pass of the host-buffer entries reads its upload arena with a kernel straight from the page-locked staging buffer and writes f / p back the same way, the caller
    polling a flag word (csrc/host_stage.hip; mm2c_tune("direct_pass", 1), the default) -- or, with the knob off, uses copy commands and a stream wait as before.  Both forms, sizes around
    the 16-byte pieces the staging kernels move (odd anchor counts: f / p end on an 8-byte boundary), one anchor, the largest pass that still takes the kernels (2^18 anchors) and the first
    that does not, many passes in a row on one context (the flag's sequence numbers), and concurrent callers."""
    import threading
    import mm2chain
    from mm2chain import params, synth
    P = params.map_ont()
    # 2 (round 6, the default): as 1, and a pass that ends in the cooperative kernel has no fourth launch -- that kernel stores f / p to the result buffer as it goes and its
    # last workgroup raises the flag; 1: the copy back as a kernel of its own (stage_out)
    mm2chain.tune("direct_pass", 1 if direct else 0); mm2chain.tune("fused_out", 1 if direct == 2 else 0)
    try:
        for n_reads, n_per, seed in [(1, 1, 1), (1, 7, 2), (3, (1, 9), 3), (5, (200, 3000), 4), (40, (50, 700), 5)]:
            off, a = _stream("mixed", n_reads, n_per, seed=700 + seed)
            f_ref, p_ref = oracle_batch(P, off, a)
            for _ in range(3):
                f, p = mm2chain.chain_batch_host(P, off, a)
                assert_same(f, p, f_ref, p_ref, off, f"direct_pass={direct}, {n_reads} reads")
        for total in ((1 << 18), (1 << 18) + 1, (1 << 18) - 1):
            off, a = _stream("mixed", 64, total // 64 + 1, seed=99)
            a = a[:total]; off = np.minimum(off, total)
            f_ref, p_ref = oracle_batch(P, off, a)
            f, p = mm2chain.chain_batch_host(P, off, a)
            assert_same(f, p, f_ref, p_ref, off, f"direct_pass={direct}, {total} anchors")
        off, a = _stream("mixed", 48, (100, 2500), seed=808)
        f_ref, p_ref = oracle_batch(P, off, a)
        errs = []

        def worker(tid):
            try:
                for rep in range(20):
                    for k in range(tid, 48, 8):
                        t = a[off[k]:off[k + 1]]
                        f, p = mm2chain.chain_task(P, t, ob.avg_qspan(t), tid=tid)
                        assert_same(f, p, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, f"read {k} thread {tid} rep {rep}")
            except Exception as e:                                  # noqa: BLE001
                errs.append(repr(e))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs[:3]
    finally:
        mm2chain.tune("direct_pass", 1); mm2chain.tune("fused_out", 1)
