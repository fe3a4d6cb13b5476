"""Long reads (round 6): batches with FEWER pieces than the GPU has wave slots -- BASELINE config 5's regime (n = 1e5 .. 1e6 anchors per task) -- and the
longest task the reference admits (chain_hardware.h:62-64: BUFFER_N = 5 187 500, refused beyond it at chain_hardware.cpp:34-37).

Which DP kernel runs is decided per run (mm2c_plan_last_route): few long pieces take sixteen waves each (chain_dp_coop: the analogue of the reference's one deep pipeline
per task, device/minimap2_opencl.cl:49,71), anything else one wave each; with long tasks cut at empty windows on the device the choice is made there.  Every case is
compared with the CPU oracle element for element."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_binding as ob
from helpers import oracle_batch, assert_same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _init():
    import mm2chain
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    mm2chain.init()
    yield
    mm2chain.shutdown()


def _stream(profile, n_reads, n_per, seed, **kw):
    from mm2chain import synth
    off, a = synth.make_stream(profile, n_reads, n_per, seed=seed, **kw)
    return off.numpy(), a.numpy().view(np.uint64)


def _plan_run(P, off, a):
    import mm2chain
    d_a = torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1, 2)).cuda()
    d_f = torch.full((a.shape[0],), -77, dtype=torch.int32, device="cuda"); d_p = torch.full_like(d_f, -77)
    plan = mm2chain.ChainPlan(P, off)
    plan.run(d_a, d_f, d_p)
    torch.cuda.synchronize()
    route, variant = plan.last_route(), plan.last_variant()
    plan.close()
    return d_f.cpu().numpy(), d_p.cpu().numpy(), route, variant


def _multi_locus(seed, n_tasks, loci, per_locus):
    """reads that hit several loci: every rid / strand change is an empty window, so the device cuts them into one piece per locus"""
    rng = np.random.default_rng(seed)
    tasks = []
    for _ in range(n_tasks):
        rows = []
        for l in range(loci):
            _, t = _stream("mixed", 1, per_locus, seed=int(rng.integers(1 << 30)))
            t = t.copy()
            t[:, 0] = (t[:, 0] & np.uint64(0xffffffff)) | (np.uint64(l + 1) << np.uint64(32))        # its own reference id
            rows.append(t)
        t = np.concatenate(rows)
        tasks.append(t[np.argsort(t[:, 0], kind="stable")])
    return tasks


def test_few_long_reads_take_sixteen_waves_each_and_many_short_ones_one():
    """the default route: 24 reads of 20 000 anchors at ava-ont density (no empty window inside: 24 pieces) -> the cooperative kernel; 6 000 reads of 300 anchors -> one
    wave each; both equal the oracle"""
    from mm2chain import params
    P = params.ava_ont()
    off, a = _stream("mixed", 24, 20000, seed=31, locus=400000)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p, route, variant = _plan_run(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"24 long reads: {variant}, route {route}")
    assert route == (24, 0, 24), (route, variant)
    P = params.map_ont()
    off, a = _stream("mixed", 6000, 300, seed=32)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p, route, variant = _plan_run(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"6 000 short reads: {variant}")
    assert route[2] == 0 and "chain_dp_tile" in variant, (route, variant)


@pytest.mark.parametrize("case", ["few-pieces", "more-pieces-than-cus", "many-pieces", "segments"])
def test_pieces_cut_on_the_device_are_routed_there(case):
    """long tasks that the device cuts at empty windows (chain_cut): 8 reads x 5 loci -> 40 pieces -> sixteen waves per piece, through the piece arrays (start / end / p base /
    avg per piece, st[] relative to the task); 60 reads x 6 loci -> 360 pieces, more than the GPU has CUs -> EIGHT waves per piece, two workgroups per CU (the count word of
    that width, chain_route); 300 reads x 9 loci -> 2 700 pieces -> one wave per piece.  `segments`: a read whose anchors carry two segment ids is
    flagged by whichever kernel met it and redone by the general variant"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    if case == "few-pieces":
        tasks = _multi_locus(5, 8, 5, 2500)
    elif case == "more-pieces-than-cus":
        tasks = _multi_locus(8, 60, 6, 1500)
    elif case == "many-pieces":
        tasks = _multi_locus(6, 300, 9, 950)
    else:
        tasks = _multi_locus(7, 6, 4, 3000)
        t = tasks[2]
        t[1000:1400, 1] |= np.uint64(1) << np.uint64(48)                                           # a second segment id inside one piece
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p, route, variant = _plan_run(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"{case}: {variant}, route {route}")
    if case == "few-pieces":
        assert route == (40, 0, 40), route
    elif case == "more-pieces-than-cus":
        assert route == (360, 0, 360), route
    elif case == "many-pieces":
        assert route[0] == 2700 and route[2] == 0, route
    else:
        assert route[2] == route[0] == 24, route
    # the same through the whole-function host entry (its chunks cut on the device as well)
    res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=0)
    for k in range(off.size - 1):
        u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, a[off[k]:off[k + 1]])
        assert np.array_equal(res[k][0], u_ref) and np.array_equal(res[k][1], b_ref), f"{case}: chains of task {k} differ"


def test_one_read_of_300000_anchors_among_short_ones():
    """one very long uncuttable read in a batch of short ones: whichever route the batch takes, f / p equal the oracle (the long read sets the tail of the batch)"""
    from mm2chain import params
    P = params.ava_ont()
    o1, a1 = _stream("mixed", 1, 300000, seed=41, locus=6000000)
    o2, a2 = _stream("mixed", 500, (200, 3000), seed=42)
    a = np.concatenate([a1, a2]); off = np.concatenate([o1, o2[1:] + o1[-1]])
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p, route, variant = _plan_run(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"{variant}, route {route}")


@pytest.mark.parametrize("preset", ["map_ont", "ava_ont", "v2"])
def test_window_starts_made_inside_the_sixteen_wave_kernel(preset):
    """a pass of few short tasks (a per-read call): the cooperative kernel makes st[] itself -- the task's x in LDS, every thread a binary search with the bounds and the
    condition of chain.c:192-193 -- instead of a prepass launch; tasks of more than 7 168 anchors keep the prepass.  Both ways against the oracle: sizes around the tile
    (64) and the cap, one target id and several (the search compares 64 bits), and the host entry a per-read call takes (run_chaining_on_hw's scalars for `v2`)"""
    import mm2chain
    from mm2chain import params
    P = {"map_ont": params.map_ont(), "ava_ont": params.ava_ont(),
         "v2": params.make_params(max_skip=2**31 - 1, max_iter=1024, q_span_override=15, flags=mm2chain.MM2C_F_IGNORE_SEG)}[preset]
    sizes = [1, 2, 63, 64, 65, 129, 700, 3000, 7167, 7168]
    parts = [_stream("mixed", 1, n, seed=80 + k) for k, n in enumerate(sizes)]
    t = np.concatenate(_multi_locus(10, 1, 4, 1200))                                               # 4 800 anchors on four target ids
    parts.append((np.array([0, t.shape[0]], np.int64), t))
    a = np.concatenate([x[1] for x in parts])
    off = np.concatenate([[0]] + [x[0][1:] + sum(y[0][-1] for y in parts[:k]) for k, x in enumerate(parts)]).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    try:
        mm2chain.tune("coop_plans", 1)
        for fuse, want in ((1, "st=kernel"), (0, "st=prepass"), (1, "st=kernel")):
            mm2chain.tune("fuse_st", fuse)
            f, p, route, variant = _plan_run(P, off, a)
            assert_same(f, p, f_ref, p_ref, off, f"{preset}, fuse_st {fuse}: {variant}")
            assert variant.startswith("chain_dp_coop<W=16") and want in variant, variant
        # one task beyond the cap in the plan: the prepass again
        o2, a2 = _stream("mixed", 1, 7169, seed=99)
        off2 = np.concatenate([off, off[-1:] + 7169]); a_2 = np.concatenate([a, a2])
        f_ref2, p_ref2 = oracle_batch(P, off2, a_2)
        f, p, route, variant = _plan_run(P, off2, a_2)
        assert_same(f, p, f_ref2, p_ref2, off2, f"{preset}, a task of 7 169 anchors: {variant}")
        assert "st=prepass" in variant, variant
        # the per-read host entry (a staged pass -> the same kernel): as ONE launch -- the kernel copies the anchors from the pinned arena itself -- and with stage_in first;
        # alternating, so that the counter of finished workgroups is handed from one kind of pass to the other
        for single in (1, 0, 1, 1, 0):
            mm2chain.tune("single_launch", single)
            for k in (3, 6, 10, 0, 9):
                t_k = a[off[k]:off[k + 1]]
                avg = ob.avg_qspan(t_k)
                f, p = mm2chain.chain_task(P, t_k, avg)
                assert_same(f, p, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None,
                            f"{preset}, single_launch {single}, chain_task of {t_k.shape[0]} anchors: {mm2chain.last_host_variant()}")
    finally:
        mm2chain.tune("coop_plans", 2); mm2chain.tune("fuse_st", 1); mm2chain.tune("single_launch", 1)


@pytest.mark.parametrize("loci", [31, 32, 33, 70])
def test_single_launch_pass_with_its_metadata_in_the_arguments_and_in_the_arena(loci):
    """a per-read call whose read hits many loci: the host cuts it into one piece per locus, and the pass runs as one launch -- up to 32 pieces with the piece offsets, avg
    and p bases in the kernel's arguments, beyond that read from the pinned arena by every workgroup.  Both sides of the line, against the oracle, and the p[] of later
    pieces (task-relative through the p base)"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    t = np.concatenate(_multi_locus(20 + loci, 1, loci, 300))
    avg = ob.avg_qspan(t)
    f_ref, p_ref, _ = ob.chain_fpv(P, t, avg)
    for single in (1, 0):
        mm2chain.tune("single_launch", single)
        try:
            for rep in range(2):
                f, p = mm2chain.chain_task(P, t, avg)
                assert_same(f, p, f_ref, p_ref, None, f"{loci} loci, single_launch {single}, run {rep}: {mm2chain.last_host_variant()}")
        finally:
            mm2chain.tune("single_launch", 1)
    assert int(p_ref.max()) > 300 * (loci - 1) - 1                                                 # (a predecessor inside the last locus: index relative to the task)
    assert "st=kernel" in mm2chain.last_host_variant(), mm2chain.last_host_variant()


def test_prepass_of_long_tasks_by_segments_and_by_task_alike():
    """plans with a task of 65 536 anchors or more run the window-start prepass with a block per 32 768 anchors of a task (the segments add up the task's sums, the last to
    arrive writes avg and the ring class); "seg_prepass" 0 keeps a block per task.  Both against the oracle, twice each (the words of the sums must be zero again after a
    run): uncuttable long reads of sizes around the segment length among short ones, and a long read of several loci (cut flags from every segment)"""
    import mm2chain
    from mm2chain import params
    P = params.ava_ont()
    parts = [_stream("mixed", 1, n, seed=60 + k, locus=20 * n) for k, n in enumerate((32768 * 2, 32768 * 3 + 1, 70001, 131072 + 255))]
    parts.append(_stream("mixed", 40, (300, 4000), seed=66))
    t = np.concatenate(_multi_locus(9, 1, 7, 12000))                                              # 84 000 anchors, seven loci
    parts.append((np.array([0, t.shape[0]], np.int64), t))
    a = np.concatenate([x[1] for x in parts])
    off = np.concatenate([[0]] + [x[0][1:] + sum(y[0][-1] for y in parts[:k]) for k, x in enumerate(parts)]).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    seen = {}
    try:
        for seg in (1, 0, 1):
            mm2chain.tune("seg_prepass", seg)
            for rep in range(2):
                f, p, route, variant = _plan_run(P, off, a)
                assert_same(f, p, f_ref, p_ref, off, f"seg_prepass {seg}, run {rep}: {variant}, route {route}")
                seen.setdefault(seg, (route, variant))
                assert (route, variant) == seen[seg]
        assert seen[0] == seen[1], seen                                                            # the same pieces, the same kernels either way
    finally:
        mm2chain.tune("seg_prepass", 1)


def test_one_task_at_the_references_buffer_limit():
    """chain_hardware.h:62-64: BUFFER_N = 332 000 000 / 2 / 32 = 5 187 500 anchors is the longest call the reference's device buffers hold (chain_hardware.cpp:34-37 refuses
    more).  One task of exactly that size through the reference's own symbol run_chaining_on_hw (V2 scalars: look-back <= 1 024, no max-skip), compared with the oracle
    element for element; then one of 2 000 000 anchors through the extended entry with the stock map-ont scalars (V1)."""
    import mm2chain
    from mm2chain import params
    BUFFER_N = 332000000 // 2 // 32
    assert BUFFER_N == 5187500
    off, a = _stream("mixed", 1, BUFFER_N, seed=51, locus=20 * BUFFER_N)                           # ava-ont density, 104 Mb locus: x stays below 2^31
    assert a.shape[0] == BUFFER_N
    avg = ob.avg_qspan(a)
    Pv2 = params.make_params(max_skip=2**31 - 1, max_iter=1024, q_span_override=15, flags=mm2chain.MM2C_F_IGNORE_SEG)
    f_ref, p_ref, _ = ob.chain_fpv(Pv2, a, avg)
    ret, f, p = mm2chain.run_chaining_on_hw(BUFFER_N, 5000, 5000, 500, 15, avg, a)
    assert ret == 0
    assert_same(f, p, f_ref, p_ref, None, f"run_chaining_on_hw, n = BUFFER_N: {mm2chain.last_host_variant()}")
    assert int(p.max()) < BUFFER_N and int(f.max()) > 10**6                                        # (a chain of millions of anchors: scores beyond 2^20)
    n2 = 2000000
    t = a[:n2]
    P = params.map_ont()
    avg2 = ob.avg_qspan(t)
    f_ref, p_ref, _ = ob.chain_fpv(P, t, avg2)
    f, p = mm2chain.chain_task(P, t, avg2)
    assert_same(f, p, f_ref, p_ref, None, f"mm2c_chain_task_host, n = 2 000 000: {mm2chain.last_host_variant()}")


def test_the_reference_symbol_refuses_a_call_beyond_the_hosts_buffer_size():
    """chain_hardware.cpp:34-37: n > BUFFER_N -> "Error: The size of the call ... exceeds buffer size" and exit(1).  The drop-in keeps that contract for the size the host
    states in hardware_init (main.c:367 passes BUFFER_N); in a child process, because the symbol ends it"""
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'minimap2-fpga_amd')!r})\n"
        "import mm2chain\n"
        "assert mm2chain.hardware_init(1000, b'none.xclbin')\n"
        "a = np.zeros((1001, 2), np.uint64); a[:, 0] = np.arange(1001) * 7 + (1 << 32); a[:, 1] = (15 << 32) | (np.arange(1001) * 7 + 20)\n"
        "ret, f, p = mm2chain.run_chaining_on_hw(1000, 5000, 5000, 500, 15, 0.15, a[:1000])\n"
        "print('accepted', ret, flush=True)\n"
        "mm2chain.run_chaining_on_hw(1001, 5000, 5000, 500, 15, 0.15, a)\n"
        "print('not reached', flush=True)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stdout, r.stderr)
    assert "accepted 0" in r.stdout and "not reached" not in r.stdout
    assert "The size of the call (n = 1001) exceeds buffer size (1000)" in r.stderr, r.stderr


def test_exit_without_shutdown_after_a_split_batch_keeps_the_exit_code():
    """advisor, round 5: the persistent workers of split batches must not turn a host's exit() into SIGABRT (a joinable std::thread destroyed by a static destructor).  A
    child lists the one GPU twice, runs a batch big enough to be split over the two device slots, and leaves through sys.exit(7) without mm2c_shutdown."""
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'minimap2-fpga_amd')!r})\n"
        "import mm2chain\n"
        "from mm2chain import params, synth\n"
        "mm2chain.init_devices([0, 0])\n"
        "mm2chain.tune('multi_min_anchors', 1000)\n"
        "off, a = synth.make_stream('mixed', 64, 2000, seed=3)\n"
        "f, p = mm2chain.chain_batch_host(params.map_ont(), off.numpy(), a.numpy().view(np.uint64))\n"
        "print('ran', int(f.shape[0]), flush=True)\n"
        "os._exit(7) if os.environ.get('HARD') else sys.exit(7)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stdout[-300:], r.stderr[-600:])
    assert "ran 128000" in r.stdout
