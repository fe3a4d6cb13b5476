"""GPU parity tests: the HIP kernel, called through the C ABI, must equal the CPU oracle bit for bit (f[] and p[]).
Reference path: chain.c:184-238 (V1) and device/minimap2_opencl.cl (V2)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle_binding as ob
from helpers import mk_anchor, pack, oracle_batch, gpu_batch, assert_same

pytestmark = pytest.mark.gpu

INT32_MAX = 2**31 - 1


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


@pytest.mark.parametrize("profile", ["sparse", "mixed", "dense", "colinear"])
def test_profiles_map_ont(profile):
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream(profile, 96, 3000, seed=11)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, profile)


def test_ragged_and_tiny_tasks():
    from mm2chain import params, synth
    P = params.map_ont()
    parts = []
    sizes = [0, 1, 2, 63, 64, 65, 127, 128, 129, 0, 500, 1, 0]
    for k, n in enumerate(sizes):
        if n:
            _, a = synth.make_stream("dense", 1, n, seed=100 + k)
            parts.append(a.numpy().view(np.uint64))
    a = np.concatenate(parts)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "ragged")


@pytest.mark.parametrize("max_skip,max_iter,gap_scale,bw", [
    (0, 5000, 1.0, 500), (3, 5000, 1.0, 500), (25, 200, 1.0, 500), (25, 50, 0.8, 500), (INT32_MAX, 1024, 1.0, 500),
    (25, 5000, 1.37, 100), (1, 64, 1.0, 2000), (25, 63, 1.0, 500), (25, 65, 1.0, 500), (INT32_MAX, 5000, 0.8, 500)])
def test_parameter_corners(max_skip, max_iter, gap_scale, bw):
    from mm2chain import params
    P = params.make_params(max_skip=max_skip, max_iter=max_iter, gap_scale=gap_scale, bw=bw)
    off, a = _stream("dense", 24, (300, 2500), seed=5)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"corner {max_skip},{max_iter},{gap_scale},{bw}")


@pytest.mark.parametrize("ring_class", [3, 4, 0, 1, 2])
def test_ring_classes_and_far_lookback(ring_class):
    """look-back far beyond the LDS ring: a very dense locus (window ~ max_iter) with the early exit mostly disabled; ring class 3 = the tile
    kernel (the default), 0 / 1 / 2 = the first-generation kernel with 256 / 512 / 1024 anchors of ring"""
    import mm2chain
    from mm2chain import params
    mm2chain.tune("ring_class", ring_class)
    try:
        for P in (params.make_params(max_skip=400, max_iter=3000), params.map_ont(), params.make_params(max_skip=INT32_MAX, max_iter=1500)):
            off, a = _stream("dense", 6, 4000, seed=21, locus=6000)
            f_ref, p_ref = oracle_batch(P, off, a)
            f, p = gpu_batch(P, off, a)
            assert_same(f, p, f_ref, p_ref, off, f"ring {ring_class}")
    finally:
        mm2chain.tune("ring_class", 3)


@pytest.mark.parametrize("max_skip", [25, 1000, INT32_MAX])
def test_window_lengths_around_ring_and_victim_boundaries(max_skip):
    """one dense cluster per task (every anchor within max_dist_x of every other): look-back lengths around every boundary of the tile
    kernel -- the tile (64), the f / p rings (128 anchors before the own tile), the x / q rings (448 before the own tile), beyond (L2/HBM)
    -- and of the first-generation kernel (256, 320/384/448), +-1 anchor each"""
    from mm2chain import params
    rng = np.random.default_rng(int(max_skip) % 1000)
    sizes = [63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319, 320, 321, 383, 384, 385, 447, 448, 449, 511, 512, 513, 575, 576, 577, 640, 900, 1500]
    tasks = []
    for n in sizes:
        pos = 50000 + np.cumsum(rng.integers(0, 3, n))           # spans < 2n bp, far below max_dist_x
        q = 100 + np.cumsum(rng.integers(0, 4, n))
        x = (np.uint64(3) << np.uint64(32)) | pos.astype(np.uint64)
        y = (np.uint64(15) << np.uint64(32)) | q.astype(np.uint64)
        tasks.append(np.stack((x, y), 1))
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    P = params.make_params(max_skip=max_skip, bw=5000)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"boundaries max_skip={max_skip}")


@pytest.mark.parametrize("max_skip,gap_scale,bw", [(25, 1.0, 500), (3, 1.0, 500), (0, 1.0, 500), (25, 0.8, 500), (25, 1.3, 600), (7, 1.0, 5000), (25, 1.0, 4999),
                                                   (-1, 1.0, 500), (-1, 0.5, 500), (25, 1.0, 0), (25, 1.0, -1)])
def test_tile_kernel_paths(max_skip, gap_scale, bw):
    """what selects the code paths of the second-generation kernel: runs of equal x shorter and longer than a tile (the hand-written scan hands
    an anchor whose x equals its predecessor's to the C++ scan; a run that crosses the tile start needs the per-lane dr != 0 test), per-anchor
    spans, windows that end inside / at / beyond the f-p rings and the x-q rings with the early exit firing at different depths, gap_scale != 1
    with and without the cost table (bw <= 511 or not), max_dq - 1 >= bw or not (the three-instruction filter needs it)"""
    from mm2chain import params
    rng = np.random.default_rng(1000 * (max_skip + 2) + bw + 2)
    tasks = []
    for n, dup, dens in [(700, 0.0, 1), (900, 0.3, 1), (1300, 0.9, 2), (2500, 0.5, 6), (4000, 0.2, 10), (130, 0.97, 1), (3000, 0.0, 20)]:
        step = np.where(rng.random(n) < dup, 0, rng.integers(1, 12 * dens + 2, n))     # dup: fraction of anchors with the x of their predecessor
        pos = (1 << 22) + np.cumsum(step)
        q = 50 + np.cumsum(np.where(rng.random(n) < 0.15, rng.integers(-400, 400, n), rng.integers(0, 14 * dens, n)))
        span = np.where(rng.random(n) < 0.5, 15, rng.integers(8, 40, n))
        x = (np.uint64(1) << np.uint64(32)) | pos.astype(np.uint64)
        y = (span.astype(np.uint64) << np.uint64(32)) | (np.maximum(q, 1).astype(np.uint64) & np.uint64(0xffffffff))
        o = np.argsort(x, kind="stable")
        tasks.append(np.stack((x[o], y[o]), 1))
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    P = params.make_params(max_skip=max_skip, gap_scale=gap_scale, bw=bw)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"tile kernel paths max_skip={max_skip} gap_scale={gap_scale} bw={bw}")
    assert bw <= 0 or int((p_ref >= 0).sum()) > a.shape[0] // 3   # (a negative max_skip: the first skip event ends a scan; bw < 0: nothing chains)


@pytest.mark.parametrize("compact,gap_scale", [(1, 0.8), (1, 1.0), (0, 0.8), (0, 1.0)])
def test_drive_every_fold_of_the_hand_written_loop(compact, gap_scale, knobs):
    """Inputs written for the rare exits of the fold (chain.c:226-233 as the hand-written loop restates it): with max_skip 1 and 3 on branching chains the
    `break` fires in every form -- in fold A, in the closed forms (B0, Lcfb) and inside the max-plus scan (Lbk) -- and with anchors 3 apart in x the windows
    (max_dist_x 5000: > 1 600 anchors) reach beyond the ring of 16 tiles, so the `far` instantiation folds chunks from memory as well.  Four instantiation
    families: compact / 32-bit ring x gap cost computed / from the table (gap_scale 0.8).  tests/test_gpu_labels.py counts the labels these (and the other
    parity inputs) reach in the real assembly; the table it printed before this test existed had four cold entries, all in the compact + table rows."""
    from helpers import fold_driver_tasks
    from mm2chain import params
    knobs("compact_ring", compact)
    rng = np.random.default_rng(4242 + 10 * compact + int(gap_scale * 10))
    for max_skip in (1, 3, 25):
        tasks = fold_driver_tasks(rng)
        a = np.concatenate(tasks)
        off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
        P = params.make_params(max_skip=max_skip, gap_scale=gap_scale)
        f_ref, p_ref = oracle_batch(P, off, a)
        v = []
        f, p = gpu_batch(P, off, a, variant=v)
        assert_same(f, p, f_ref, p_ref, off, f"fold drivers, max_skip={max_skip} gap_scale={gap_scale} compact={compact}: {v[0]}")
        assert "loop=asm" in v[0] and f"compact={compact}" in v[0] and f"TAB={int(gap_scale != 1.0)}" in v[0], v


def _steep_colinear_task(rng, n, step_lo, step_hi, jitter_lo, jitter_hi):
    """one colinear chain whose links are `step` apart in x and differ by `jitter` between x and q: every anchor's best predecessor is its neighbour, the gap cost of a
    link is (int)(jitter * avg) + log2(jitter) / 2"""
    step = rng.integers(step_lo, step_hi + 1, n)
    jit = rng.integers(jitter_lo, jitter_hi + 1, n) * rng.choice([-1, 1], n)
    pos = (1 << 20) + np.cumsum(step)
    q = 100 + np.cumsum(np.maximum(step + jit, 1))
    assert pos[-1] < (1 << 31) and q[-1] < (1 << 31)
    x = (np.uint64(1) << np.uint64(32)) | pos.astype(np.uint64)
    y = (np.uint64(15) << np.uint64(32)) | q.astype(np.uint64)
    return np.stack((x, y), 1)


@pytest.mark.parametrize("case", ["profiles", "fold-drivers", "tile-paths", "compact-corners", "random-scalars", "ava-ont", "tiny-and-ragged", "steep-scores"])
def test_several_waves_per_task_equal_the_oracle(case, knobs):
    """chain_dp_coop (csrc/chain_dp_coop.h): a workgroup of 16 waves per task.  Candidates are counted and the older tiles reduced in parallel; an anchor with at
    most max_skip candidates in its window cannot take the `break` of chain.c:231, so its result is the plain maximum (nearest j on ties, chain.c:226); every
    other anchor takes the exact scan (hand-written loop or C++).  Both kinds, the hand-over between them inside a tile, windows beyond the ring, equal-x runs,
    per-anchor spans, max_skip from -1 to INT_MAX, the gap-cost table: same f / p as the oracle."""
    from helpers import fold_driver_tasks, respan_q
    from mm2chain import params, synth
    knobs("coop_plans", 1)
    rng = np.random.default_rng(515)
    runs = []
    if case == "profiles":
        for prof in ("sparse", "mixed", "dense", "colinear"):
            off, a = _stream(prof, 6, (200, 3000), seed=61)
            runs += [(params.map_ont(), off, a), (params.make_params(max_skip=INT32_MAX, max_iter=1024), off, a)]
    elif case == "fold-drivers":
        tasks = fold_driver_tasks(rng)
        for ms, gs in ((1, 1.0), (3, 0.8), (25, 1.0), (0, 1.0), (-1, 1.0), (300, 1.0)):
            runs.append((params.make_params(max_skip=ms, gap_scale=gs), None, tasks))
    elif case == "tile-paths":
        tasks = []
        for n, dup, dens in [(700, 0.0, 1), (900, 0.3, 1), (1300, 0.9, 2), (2500, 0.5, 6), (4000, 0.2, 10), (130, 0.97, 1), (3000, 0.0, 20)]:
            step = np.where(rng.random(n) < dup, 0, rng.integers(1, 12 * dens + 2, n))
            pos = (1 << 22) + np.cumsum(step)
            q = 50 + np.cumsum(np.where(rng.random(n) < 0.15, rng.integers(-400, 400, n), rng.integers(0, 14 * dens, n)))
            span = np.where(rng.random(n) < 0.5, 15, rng.integers(8, 40, n))
            x = (np.uint64(1) << np.uint64(32)) | pos.astype(np.uint64)
            y = (span.astype(np.uint64) << np.uint64(32)) | (np.maximum(q, 1).astype(np.uint64) & np.uint64(0xffffffff))
            o = np.argsort(x, kind="stable")
            tasks.append(np.stack((x[o], y[o]), 1))
        for ms, gs, bw in ((25, 1.0, 500), (3, 1.0, 500), (25, 0.8, 500), (7, 1.0, 5000), (25, 1.0, 0), (25, 1.3, 600), (25, 1.0, -1), (1000, 1.0, 500)):
            runs.append((params.make_params(max_skip=ms, gap_scale=gs, bw=bw), None, tasks))
    elif case == "compact-corners":
        P = params.map_ont()
        tasks = []
        for k, (prof, n, locus) in enumerate([("mixed", 3000, None), ("dense", 4000, 20000), ("colinear", 2500, None), ("mixed", 1, None)]):
            base = synth.make_stream(prof, 1, n, seed=770 + k, locus=locus)[1].numpy().view(np.uint64)
            tasks += [respan_q(rng, base, 5000, mode) for mode in (0, 2, 4, 7)]
        runs.append((P, None, tasks))
    elif case == "random-scalars":
        for seed in range(8):
            r2 = np.random.default_rng(2000 + seed)
            P = params.make_params(max_dist_x=int(r2.choice([50, 700, 5000, 100000])), max_dist_y=int(r2.choice([60, 700, 5000])), bw=int(r2.choice([0, 10, 500, 5000])),
                                   max_skip=int(r2.choice([-1, 0, 1, 5, 25, 300, INT32_MAX])), max_iter=int(r2.choice([1, 63, 64, 65, 200, 1024, 5000, INT32_MAX])),
                                   gap_scale=float(r2.choice([1.0, 1.0, 0.5, 2.25])))
            runs.append((P, None, [_random_task(r2, int(r2.integers(1, 1500)), int(r2.integers(1, 4)), 1, bool(r2.integers(0, 2))) for _ in range(8)]))
    elif case == "ava-ont":
        off, a = _stream("mixed", 2, 6000, seed=62, locus=120000)
        runs.append((params.ava_ont(), off, a))
        off, a = _stream("dense", 2, 5000, seed=63, locus=9000)           # windows of 1 400 anchors: beyond the ring of 16 tiles
        runs += [(params.map_ont(), off, a), (params.make_params(max_skip=INT32_MAX, max_iter=1024), off, a)]
    elif case == "steep-scores":
        # round 4's advisor: the straight-line pushes carry score << 7 | origin in ONE word, right only while a chain gains at most 255 per link.  A negative
        # gap_scale turns the gap cost into a gain (chain.c:219): -15 * ((int)(300 * .15) + 4) = +735 on top of the span per link, 30 000 links: f passes 2^23
        t = _steep_colinear_task(rng, 30000, 20, 60, 200, 420)
        runs.append((params.make_params(gap_scale=-15.0, bw=500), None, [t, t[:700]]))
    else:
        sizes = [1, 2, 63, 64, 65, 127, 128, 129, 1000, 0, 5, 4097]
        tasks = [synth.make_stream("mixed", 1, max(n, 1), seed=900 + k)[1].numpy().view(np.uint64)[:n] for k, n in enumerate(sizes)]
        runs.append((params.map_ont(), None, tasks))
    for P, off, a in runs:
        if off is None:
            off = np.concatenate(([0], np.cumsum([t.shape[0] for t in a]))).astype(np.int64)
            a = np.concatenate(a)
        f_ref, p_ref = oracle_batch(P, off, a)
        v = []
        f, p = gpu_batch(P, off, a, variant=v)
        assert_same(f, p, f_ref, p_ref, off, f"several waves per task, {case}, {params.as_dict(P)}: {v[0]}")
        if case == "steep-scores":
            assert int(f_ref.max()) > (1 << 23), "the task was meant to carry scores beyond the one-word key"
        simple = P.gap_scale == 1.0 or P.bw <= 511
        assert v[0].startswith("chain_dp_coop<W=16") == (simple and P.bw >= 0 and min(P.max_dist_x, P.max_dist_y) - 1 >= P.bw), (v, params.as_dict(P))


@pytest.mark.parametrize("max_skip,far_ring", [(25, 1), (1000, 1), (1000, 0), (25, 2), (INT32_MAX, 1)])
def test_far_lookback_in_partial_tail_tiles(max_skip, far_ring, knobs):
    """the last tile of a task holds cnt < 64 anchors; when their windows reach beyond the LDS ring the `far` instantiation of the hand-written
    loop runs with lanes that hold no anchor, prefetches of x / q beyond the ring in flight at the tile's (and the task's) end, and far stamps
    for a partial tile (the memory fault of round 2 came from that corner: loads still in flight when the block ended).  Tasks of every
    residue class of interest, every anchor inside one window, so that scans really go beyond the ring unless the break comes first."""
    from mm2chain import params, synth
    knobs("far_ring", far_ring)
    sizes = [449, 450, 511, 513, 575, 577, 640 + 1, 960 + 63, 1024 + 1, 1024 + 31, 1087, 1500 + 37, 2048 + 2, 3000 + 5, 64 * 40 + 33, 7 * 64 + 62]
    parts = [synth.make_stream("dense", 1, n, seed=900 + k, locus=3000 if k % 2 else 4500)[1].numpy().view(np.uint64) for k, n in enumerate(sizes)]
    a = np.concatenate(parts)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    P = params.make_params(max_skip=max_skip)
    f_ref, p_ref = oracle_batch(P, off, a)
    v = []
    f, p = gpu_batch(P, off, a, variant=v)
    assert_same(f, p, f_ref, p_ref, off, f"partial tail tiles, max_skip={max_skip}, far_ring={far_ring}: {v[0]}")
    assert "FAR=1" in v[0] and "loop=asm" in v[0], v          # (max_skip >= max_iter too: it runs with max_skip = max_iter - 1)


@pytest.mark.parametrize("far_ring", [1, 2, 0])
def test_ring_size_classes_in_one_batch(far_ring):
    """plans give tasks whose scans are expected to go far beyond the 448-anchor LDS ring an instantiation with a ring twice as long (chosen per task
    by the prepass; far_ring 2: every task, 0: none): a batch under ava-ont scalars with tasks of both kinds, long ones that are cut into pieces
    on the device included, must come out the same whichever instantiation ran"""
    import mm2chain
    from mm2chain import params, synth
    P = params.ava_ont()
    parts = []
    for prof, n, locus, seed in [("mixed", 12000, 400000, 1), ("colinear", 9000, 400000, 2), ("mixed", 700, None, 3), ("dense", 5000, 30000, 4),
                                 ("mixed", 20000, 400000, 5), ("sparse", 3000, None, 6), ("mixed", 1023, 400000, 7)]:
        parts.append(synth.make_stream(prof, 1, n, seed=400 + seed, locus=locus)[1].numpy().view(np.uint64))
    a = np.concatenate(parts)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in parts]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    mm2chain.tune("far_ring", far_ring)
    try:
        f, p = gpu_batch(P, off, a)
    finally:
        mm2chain.tune("far_ring", 1)
    assert_same(f, p, f_ref, p_ref, off, f"far_ring={far_ring}")


@pytest.mark.parametrize("q24,gap_scale", [(1, 1.0), (0, 1.0), (1, 0.8)])
def test_long_ring_in_the_q24_form(q24, gap_scale, knobs):
    """Round 5: the long ring (ring-size class 1) keeps the low 16 bits of x and 24 bits of q per anchor -- a 4-byte slot and one byte of a ring of its own, 7 KB for
    16 tiles instead of 10 -- exact for tasks whose q values are below 2^24; the prepass keeps any other task out of the class.  Every task in the long ring
    (far_ring 2, compact ring off) under ava-ont scalars: q spread over 400 kb (differences that alias mod 2^16 in every tile), q moved up to just below 2^24 and
    to 2^24 and beyond (those tasks must take the 32-bit short ring and still be right), multiples of 65536 added to random anchors, windows longer than the ring
    (the `far` forms), equal-x runs, the gap-cost table (gap_scale 0.8 with bw <= 511)."""
    from helpers import respan_q
    from mm2chain import params, synth
    P = params.make_params(max_dist_x=10000, max_dist_y=10000, bw=2000 if gap_scale == 1.0 else 500, gap_scale=gap_scale)
    rng = np.random.default_rng(2424)
    knobs("compact_ring", 0); knobs("far_ring", 2); knobs("q24_ring", q24); knobs("plan_cut_min", 15000)
    tasks = []
    for k, (prof, n, locus) in enumerate([("mixed", 6000, 400000), ("dense", 5000, 30000), ("colinear", 3000, 400000), ("mixed", 20000, 400000), ("dense", 2500, 9000),
                                          ("mixed", 700, None), ("mixed", 1, None), ("mixed", 65, 4000)]):
        base = synth.make_stream(prof, 1, n, seed=2400 + k, locus=locus)[1].numpy().view(np.uint64)
        tasks.append(base)
        if n <= 6000:
            tasks += [respan_q(rng, base, 10000, mode) for mode in (2, 7)]                      # multiples of 65536 up to 2^24 added to random anchors
            qmax = int((base[:, 1] & np.uint64(0xffffffff)).max()) if n else 0
            for top in ((1 << 24) - 1, 1 << 24, (1 << 24) + 70000, (1 << 31) + 5):              # the largest q of the task: the last value the q24 ring holds, and beyond
                t = base.copy()
                t[:, 1] = (t[:, 1] & np.uint64(0xffffffff00000000)) | ((t[:, 1] & np.uint64(0xffffffff)) + np.uint64(top - qmax))
                tasks.append(t)
    # runs of equal x inside long windows
    t = synth.make_stream("dense", 1, 4000, seed=2499, locus=20000)[1].numpy().view(np.uint64).copy()
    t[1::3, 0] = t[0:-1:3, 0][: t[1::3, 0].shape[0]]
    tasks.append(t[np.argsort(t[:, 0], kind="stable")])
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([x.shape[0] for x in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    v = []
    f, p = gpu_batch(P, off, a, variant=v)
    assert_same(f, p, f_ref, p_ref, off, f"long ring, q24={q24}, gap_scale={gap_scale}: {v[0]}")
    assert "loop=asm" in v[0] and "classes=1" in v[0] and f"q24={q24}" in v[0] and f"TAB={int(gap_scale != 1.0)}" in v[0], v


@pytest.mark.parametrize("preset,compact,wide_pct", [("map-ont", 1, 100), ("map-ont", 1, 40), ("map-ont", 1, 0), ("map-ont", 0, 40), ("ava-ont", 1, 100), ("asm20", 1, 100)])
def test_compact_ring_takes_the_tasks_whose_q_values_allow_it(preset, compact, wide_pct, knobs):
    """The tile kernel keeps the LOW 16 BITS of x and q of the ring anchors (4 bytes per anchor, Lds<..., C16>): exact while max_dist_x < 2^16 and the
    task's q values span at most 65535 - max_dq; the prepass (chain_window_start) sends every other task to the instantiations with the 32-bit ring.
    One batch with tasks on both sides of that bound and at it: q shifted by large constants (low halves wrap), multiples of 65536 added to random
    anchors (differences that alias mod 2^16), a span of exactly 65535 - max_dq and one more, q spread over 60 000, tasks long enough to be cut into
    pieces on the device (pieces inherit the class).  wide_share_threshold: the share of the batch's anchors in tasks with the 32-bit ring from which
    every task takes it (100: the split always stands, 0: never)."""
    from helpers import respan_q
    from mm2chain import params, synth
    P = {"map-ont": params.map_ont, "ava-ont": params.ava_ont, "asm20": params.asm20}[preset]()
    max_dq = min(P.max_dist_x, P.max_dist_y)
    rng = np.random.default_rng(77)
    knobs("compact_ring", compact)
    knobs("wide_share_threshold", wide_pct)
    knobs("split_streams", 2 if wide_pct == 100 else 1 if wide_pct == 40 else 0)   # the two kinds side by side on two streams: always / for batches of mixed task sizes / never
    knobs("plan_cut_min", 6000)
    tasks = []
    for k, (prof, n, locus) in enumerate([("mixed", 3000, None), ("dense", 4000, 20000), ("mixed", 9000, 300000), ("colinear", 2500, None), ("dense", 7000, 30000),
                                          ("mixed", 700, None), ("sparse", 500, None), ("mixed", 1, None), ("dense", 1500, 3000)]):
        base = synth.make_stream(prof, 1, n, seed=880 + k, locus=locus)[1].numpy().view(np.uint64)
        for mode in ((0, 1, 2, 3, 4, 5, 6, 7) if n <= 4000 else (0, 2, 5, 7)):
            tasks.append(respan_q(rng, base, max_dq, mode))
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    v = []
    f, p = gpu_batch(P, off, a, variant=v)
    assert_same(f, p, f_ref, p_ref, off, f"{preset}, compact_ring={compact}, wide_share_threshold={wide_pct}: {v[0]}")
    assert f"compact={compact}" in v[0] and "loop=asm" in v[0], v


@pytest.mark.parametrize("entry", ["task", "batch", "mm_chain_dp_batch", "task-several-waves", "batch-several-waves"])
def test_host_buffer_entries_take_the_compact_ring(entry, knobs):
    """The entries that move anchors and f / p (or chains) across PCIe -- mm2c_chain_task_host (the extended run_chaining_on_hw, chain.c:103),
    mm2c_chain_batch_host, its chunked pipeline and mm2c_mm_chain_dp_batch_host -- give their passes the prepass classes as plans do, so the
    tasks whose q values allow it run the compact-ring instantiation there too (round 3: these entries launched without the class array and kept
    the 32-bit rings).  Same corner tasks as the plan test: q shifted, aliased mod 2^16, spans at and one beyond the bound."""
    import mm2chain
    from helpers import respan_q
    from mm2chain import params, synth
    P = params.map_ont()
    max_dq = min(P.max_dist_x, P.max_dist_y)
    rng = np.random.default_rng(78)
    knobs("wide_share_threshold", 100)                       # the split between compact and 32-bit tasks always stands
    coop = entry.endswith("several-waves")                   # a pass of few pieces takes chain_dp_coop by default; with it switched off, one wave per piece and the compact ring
    knobs("coop_waves", 8 if coop else 0)
    entry = entry.replace("-several-waves", "")
    want = ("chain_dp_coop<W=16", "coop=16") if coop else ("compact=1", "loop=asm")
    tasks = []
    for k, (prof, n, locus) in enumerate([("mixed", 3000, None), ("dense", 4000, 20000), ("colinear", 2500, None), ("mixed", 700, None), ("sparse", 500, None), ("mixed", 1, None)]):
        base = synth.make_stream(prof, 1, n, seed=990 + k, locus=locus)[1].numpy().view(np.uint64)
        for mode in (0, 1, 2, 3, 4, 5, 7):
            tasks.append(respan_q(rng, base, max_dq, mode))
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    if entry == "task":
        for k in range(len(tasks)):
            t = tasks[k]
            f1, p1 = mm2chain.chain_task(P, t, ob.avg_qspan(t), tid=k)
            assert_same(f1, p1, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, f"task host, task {k}")
            assert all(w in mm2chain.last_host_variant() for w in want), mm2chain.last_host_variant()
    elif entry == "batch":
        f, p = mm2chain.chain_batch_host(P, off, a)
        assert_same(f, p, f_ref, p_ref, off, entry)
        assert all(w in mm2chain.last_host_variant() for w in want), mm2chain.last_host_variant()
    else:
        res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a)
        _assert_chains(res, P, 3, 40, off, a, "whole-function batch entry with the compact ring")
        assert "compact=1" in mm2chain.last_host_variant(), mm2chain.last_host_variant()


def test_ava_ont_and_asm20_shapes():
    from mm2chain import params
    for P, span in ((params.ava_ont(), 15), (params.asm20(), 19)):
        off, a = _stream("mixed", 16, (2000, 6000), seed=9, q_span=span)
        f_ref, p_ref = oracle_batch(P, off, a)
        f, p = gpu_batch(P, off, a)
        assert_same(f, p, f_ref, p_ref, off, f"span {span}")


@pytest.mark.parametrize("preset,profile", [("asm20", "mixed"), ("asm20", "dense"), ("ava-ont", "mixed"), ("ava-ont", "colinear"), ("map-ont", "ragged")])
def test_configs_4_and_5_at_their_bench_shapes(preset, profile):
    """BASELINE configs 4 and 5 (stand-ins of SURVEY 8d) at the shapes `bench.py --preset asm20 / ava-ont` times, and the ragged variant of
    config 2: 256 reads of 7 500 anchors with span 19 (asm20: options.c:113-122), 256 reads of 20 000 anchors in a 400 kb locus under
    bw 2000 / max_gap 10000 (ava-ont: options.c:83-86), 256 reads of U[1000, 9000] anchors; the same generator and seed as bench.py, so these
    are the first reads of the benchmarked batches.  HIP against the oracle, element by element."""
    from mm2chain import params, synth
    n_reads = 256
    if preset == "asm20":
        P, (off, a) = params.asm20(), synth.make_stream(profile, n_reads, 7500, seed=20240, q_span=19)
    elif preset == "ava-ont":
        P, (off, a) = params.ava_ont(), synth.make_stream(profile, n_reads, 20000, seed=20240, locus=400000)
    else:
        P, (off, a) = params.map_ont(), synth.make_stream("mixed", n_reads, (1000, 9000), seed=20240)
    off, a = off.numpy(), a.numpy().view(np.uint64)
    f_ref, p_ref, _ = ob.chain_batch(P, off, a, n_threads=min(16, os.cpu_count() or 4))
    v = []
    f, p = gpu_batch(P, off, a, variant=v)
    assert_same(f, p, f_ref, p_ref, off, f"{preset} {profile}: {v[0]}")
    assert "loop=asm" in v[0] and "SKIP=1" in v[0] and "FAR=1" in v[0], v


def _multiseg_task(rng, n, n_segs):
    rows = []
    pos = 1 << 20
    q = 100
    for _ in range(n):
        pos += int(rng.integers(0, 40))          # includes dr == 0 between segments
        q += int(rng.integers(-30, 60))
        rows.append(mk_anchor(0, 3, pos, max(q, 1), span=int(rng.integers(10, 30)), seg=int(rng.integers(0, n_segs))))
    return pack(rows)


@pytest.fixture(params=[3, 4], ids=["general-in-wave-kernel", "general-in-tile-kernel"])
def general_kernel(request):
    """the segment / cDNA variant runs in the first-generation kernel by default (ring_class 3) and in the tile kernel on request (4)"""
    import mm2chain
    mm2chain.tune("ring_class", request.param)
    yield request.param
    mm2chain.tune("ring_class", 3)


@pytest.mark.parametrize("is_cdna,n_segs", [(0, 2), (1, 1), (1, 2), (0, 3)])
def test_general_variant_segments_and_cdna(is_cdna, n_segs, general_kernel):
    """chain.c:206,211-217: multi-segment (sr) and cDNA branches, non-uniform spans"""
    from mm2chain import params
    rng = np.random.default_rng(7 + is_cdna * 10 + n_segs)
    P = params.make_params(max_dist_x=800, max_dist_y=800, bw=300, is_cdna=is_cdna, n_segs=n_segs, max_skip=5)
    tasks = [_multiseg_task(rng, int(rng.integers(50, 900)), n_segs) for _ in range(12)]
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "general")


def test_simple_variant_detects_foreign_segment_ids(general_kernel):
    """n_segs == 1 but anchors carry different segment ids: the simple kernel flags the task, the general one redoes it"""
    from mm2chain import params
    rng = np.random.default_rng(3)
    P = params.make_params(max_dist_x=800, max_dist_y=800, bw=300)
    tasks = [_multiseg_task(rng, 400, 1), _multiseg_task(rng, 700, 2), _multiseg_task(rng, 300, 1)]
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "seg detect")


def test_strand_and_reference_boundaries_inside_a_task():
    from mm2chain import params
    rows = []
    for strand in (0, 1):
        for rid in (0, 1, 5):
            for k in range(150):
                rows.append(mk_anchor(strand, rid, 1000 + 11 * k, 50 + 11 * k + (k % 3)))
    # anchors at the very start of a reference and identical x
    rows += [mk_anchor(0, 7, 0, 20), mk_anchor(0, 7, 0, 40), mk_anchor(0, 7, 5, 60), mk_anchor(0, 7, 5, 60)]
    a = pack(rows)
    off = np.array([0, a.shape[0]], dtype=np.int64)
    P = params.map_ont()
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "boundaries")


def test_avg_qspan_in_kernel_matches_host_value():
    """spans non-uniform and a sum above 2^24 so that the u64 -> f32 rounding of chain.c:49 matters"""
    from mm2chain import params
    rng = np.random.default_rng(1)
    n = 120000
    pos = np.cumsum(rng.integers(1, 30, n)) + (1 << 20)
    q = np.cumsum(rng.integers(1, 30, n)) + 50
    span = rng.integers(100, 256, n)
    x = (np.uint64(2) << np.uint64(32)) | pos.astype(np.uint64)
    y = (span.astype(np.uint64) << np.uint64(32)) | q.astype(np.uint64)
    a = np.stack((x, y), 1)
    off = np.array([0, n], dtype=np.int64)
    P = params.map_ont()
    f_ref, p_ref = oracle_batch(P, off, a)
    f1, p1 = gpu_batch(P, off, a)                                        # avg computed in the kernel
    f2, p2 = gpu_batch(P, off, a, avg=[ob.avg_qspan(a)])                # avg handed in
    assert_same(f1, p1, f_ref, p_ref, off, "avg in kernel")
    assert_same(f2, p2, f_ref, p_ref, off, "avg from host")


def test_fpga_v2_through_the_reference_symbol():
    """run_chaining_on_hw (chain_hardware.h:68) == literal emulation of the .cl kernel == oracle V1(max_skip=inf, max_iter=1024)"""
    import mm2chain
    from mm2chain import params
    off, a = _stream("dense", 3, 3000, seed=17, locus=9000)
    for k in range(3):
        t = a[off[k]:off[k + 1]]
        avg = ob.avg_qspan(t)
        ns, tot, _ = ob.predict(t, 5000)
        ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 15, avg, t, ns, tot, tid=k)
        assert ret == 0
        f_lit, p_lit = ob.chain_hw_literal(5000, 5000, 500, 15, avg, t)
        assert_same(f, p, f_lit, p_lit, None, "v2 literal")
        f_v1, p_v1, _ = ob.chain_fpv(params.make_params(max_skip=INT32_MAX, max_iter=1024), t, avg)
        assert_same(f, p, f_v1, p_v1, None, "v2 as v1")


def test_reference_symbol_with_a_span_beyond_255():
    """run_chaining_on_hw takes q_span as an int (chain_hardware.h:68) and nothing bounds it: with q_span 5000 a link gains up to 5000 and a 30 000-anchor chain
    carries scores beyond 2^23 -- the lone call runs the cooperative kernel, whose one-word keys (score << 7 | origin) must not be taken then (round 4's advisor).
    Checked against the literal emulation of the .cl kernel."""
    import mm2chain
    rng = np.random.default_rng(77)
    t = _steep_colinear_task(rng, 30000, 1500, 4000, 0, 40)
    avg = 50.0
    ns, tot, _ = ob.predict(t, 5000)
    ret, f, p = mm2chain.run_chaining_on_hw(t.shape[0], 5000, 5000, 500, 5000, avg, t, ns, tot, tid=0)
    assert ret == 0
    f_lit, p_lit = ob.chain_hw_literal(5000, 5000, 500, 5000, avg, t)
    assert int(f_lit.max()) > (1 << 23)
    assert_same(f, p, f_lit, p_lit, None, "q_span 5000 through the reference symbol")
    assert mm2chain.last_host_variant().startswith("chain_dp_coop<W=16"), mm2chain.last_host_variant()


def test_host_paths_cut_tasks_at_empty_windows():
    """multi-locus reads (many references / far-apart loci): the host-buffer paths split a task into independent pieces;
    results must not depend on the piece size, incl. look-back beyond the LDS ring inside a piece"""
    import mm2chain
    from mm2chain import params
    rng = np.random.default_rng(99)
    tasks = []
    for _ in range(6):
        rows = []
        for _ in range(int(rng.integers(3, 30))):
            strand, rid = int(rng.integers(0, 2)), int(rng.integers(0, 3))
            pos = int(rng.integers(0, 1 << 26)); q = int(rng.integers(0, 3000))
            for _ in range(int(rng.integers(1, 700))):
                pos += int(rng.integers(1, 9)); q += int(rng.integers(-3, 12))
                rows.append(mk_anchor(strand, rid, pos, max(q, 0)))
        tasks.append(pack(rows))
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    P = params.make_params(max_skip=300)           # long scans: windows of up to ~1200 anchors, beyond the ring
    f_ref, p_ref = oracle_batch(P, off, a)
    try:
        for seg_min in (0, 1, 64, 1000):
            mm2chain.tune("seg_min", seg_min)
            f, p = mm2chain.chain_batch_host(P, off, a)
            assert_same(f, p, f_ref, p_ref, off, f"seg_min {seg_min}")
            f1, p1 = mm2chain.chain_task(P, tasks[0], ob.avg_qspan(tasks[0]))
            assert_same(f1, p1, f_ref[: off[1]], p_ref[: off[1]], None, f"task seg_min {seg_min}")
    finally:
        mm2chain.tune("seg_min", 256)


def _multi_locus_tasks(seed, n_tasks, loci=(3, 30), per_locus=(1, 700), seg_ids=False):
    rng = np.random.default_rng(seed)
    tasks = []
    for _ in range(n_tasks):
        rows = []
        for _ in range(int(rng.integers(*loci))):
            strand, rid = int(rng.integers(0, 2)), int(rng.integers(0, 3))
            pos = int(rng.integers(0, 1 << 26)); q = int(rng.integers(0, 3000))
            for _ in range(int(rng.integers(*per_locus))):
                pos += int(rng.integers(1, 9)); q += int(rng.integers(-3, 12))
                rows.append(mk_anchor(strand, rid, pos, max(q, 0), seg=int(rng.integers(0, 2)) if seg_ids else 0))
        tasks.append(pack(rows))
    return tasks


@pytest.mark.parametrize("seg_min,cut_min", [(1, 1), (64, 300), (256, 2000), (1000, 1)])
def test_plans_cut_long_tasks_on_the_device(seg_min, cut_min):
    """device-resident plans cut tasks of plan_cut_min anchors or more at empty windows into pieces (chain_cut); f/p, the chains of the
    device epilogue, the avg_qspan of cut and uncut tasks, and the general-variant redo pass must not notice"""
    import mm2chain
    from mm2chain import params
    tasks = _multi_locus_tasks(123, 9) + [_stream("mixed", 1, (700, 700), seed=5)[1], np.zeros((0, 2), np.uint64)] + _multi_locus_tasks(124, 3, seg_ids=True)
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    try:
        mm2chain.tune("seg_min", seg_min); mm2chain.tune("plan_cut_min", cut_min)
        for P in (params.map_ont(), params.make_params(max_skip=300), params.make_params(n_segs=2)):
            f_ref, p_ref = oracle_batch(P, off, a)
            f, p = gpu_batch(P, off, a)
            assert_same(f, p, f_ref, p_ref, off, f"seg_min {seg_min} cut_min {cut_min}")
        P = params.map_ont()
        plan = mm2chain.ChainPlan(P, off)
        d_a = torch.from_numpy(a.view(np.int64)).cuda()
        d_f = torch.empty(a.shape[0], dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
        plan.run(d_a, d_f, d_p)
        u_off, u, b_off, b = plan.chains(d_a, d_f, d_p, 3, 40)
        torch.cuda.synchronize()
        uo, bo = u_off.cpu().numpy(), b_off.cpu().numpy()
        u, b = u.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64)
        plan.close()
        for k in range(off.size - 1):
            u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, a[off[k]:off[k + 1]])
            assert np.array_equal(u[uo[k]:uo[k + 1]], u_ref) and np.array_equal(b[bo[k]:bo[k + 1]], b_ref), f"task {k}: chains differ"
    finally:
        mm2chain.tune("seg_min", 256); mm2chain.tune("plan_cut_min", 8192)


def test_plan_with_one_very_long_task_is_cut_by_default():
    """default knobs: a 60 000-anchor read made of many loci among short reads"""
    import mm2chain
    from mm2chain import params
    tasks = _multi_locus_tasks(7, 1, loci=(150, 151), per_locus=(300, 500)) + _multi_locus_tasks(8, 6)
    assert tasks[0].shape[0] > 45000
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    P = params.map_ont()
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "long task")
    # the whole-function host entry cuts on the device as well
    res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=0)
    _assert_chains(res, P, 3, 40, off, a, "long task, whole function")


def test_concurrent_callers_are_combined_into_shared_passes():
    """the reference's call pattern (map.c:561): many host threads, each blocking in a per-read chaining call"""
    import threading
    import mm2chain
    from mm2chain import params, _native as N
    P = params.map_ont()
    off, a = _stream("mixed", 96, (50, 1500), seed=61)
    f_ref, p_ref = oracle_batch(P, off, a)
    results = {}

    def worker(tid):
        for k in range(tid, 96, 12):
            t = a[off[k]:off[k + 1]]
            results[k] = mm2chain.chain_task(P, t, ob.avg_qspan(t), tid=tid)

    st0 = N.Stats(); N.load().mm2c_get_stats(st0)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
    for x in th: x.start()
    for x in th: x.join()
    st1 = N.Stats(); N.load().mm2c_get_stats(st1)
    for k in range(96):
        assert_same(results[k][0], results[k][1], f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, f"thread task {k}")
    assert st1.tasks - st0.tasks == 96 and 1 <= st1.passes - st0.passes <= 96


def test_prediction_pass_matches_chain_c_53_78():
    import mm2chain
    from mm2chain import params
    off, a = _stream("dense", 7, (100, 4000), seed=41, locus=7000)
    plan = mm2chain.ChainPlan(params.map_ont(), off)
    ns, ts, tt = plan.predict(torch.from_numpy(a.view(np.int64)).cuda())
    torch.cuda.synchronize()
    ns, ts, tt = ns.cpu().numpy(), ts.cpu().numpy(), tt.cpu().numpy()
    for k in range(7):
        ns_ref, tot_ref, trip_ref = ob.predict(a[off[k]:off[k + 1]], 5000)
        assert np.array_equal(ns[off[k]:off[k + 1]], ns_ref) and ts[k] == tot_ref and tt[k] == trip_ref
    plan.close()


def test_host_paths_and_mm_chain_dp():
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream("mixed", 5, (500, 3000), seed=31)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = mm2chain.chain_batch_host(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "batch host")
    for k in range(5):
        t = a[off[k]:off[k + 1]]
        f1, p1 = mm2chain.chain_task(P, t, ob.avg_qspan(t), tid=k)
        assert_same(f1, p1, f_ref[off[k]:off[k + 1]], p_ref[off[k]:off[k + 1]], None, "task host")
        u, b = mm2chain.mm_chain_dp(5000, 5000, 500, 25, 5000, 3, 40, 1.0, 0, 1, t)
        u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, t)
        assert np.array_equal(u, u_ref) and np.array_equal(b, b_ref), "mm_chain_dp chains differ"
    u, b = mm2chain.mm_chain_dp(5000, 5000, 500, 25, 5000, 3, 40, 1.0, 0, 1, np.zeros((0, 2), np.uint64))
    assert u.size == 0 and b.size == 0


def _mixed_batch_with_gaps():
    """40 mixed reads, 10 sparse reads without any chain and three empty tasks in between"""
    off0, a0 = _stream("mixed", 40, (1, 4000), seed=77)
    off1, a1 = _stream("sparse", 10, (1, 30), seed=78)
    sizes = np.concatenate([np.diff(off0)[:20], [0], np.diff(off1), [0, 0], np.diff(off0)[20:]])
    a = np.concatenate([a0[:off0[20]], a1, a0[off0[20]:]])
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64), a


def _assert_chains(res, P, min_cnt, min_sc, off, a, what):
    n_chains = 0
    for k, (u, b) in enumerate(res):
        u_ref, b_ref = ob.mm_chain_dp(P, min_cnt, min_sc, a[off[k]:off[k + 1]])
        assert np.array_equal(u, u_ref), f"{what}: task {k}: u differs ({u.size} vs {u_ref.size} chains)"
        assert np.array_equal(b, b_ref), f"{what}: task {k}: b differs"
        n_chains += u.size
    return n_chains


@pytest.fixture(params=[1, 0], ids=["epi-lds", "epi-hbm"])
def epi_path(request):
    """the two forms of the device epilogue: tasks that fit the LDS in the fused kernel (default), or kernels A / B / C for every task"""
    import mm2chain
    mm2chain.tune("epi_fused", request.param)
    yield request.param
    mm2chain.tune("epi_fused", 1)


@pytest.mark.parametrize("epilogue_threads", [0, 1, 5])
def test_batched_mm_chain_dp_equals_task_by_task_calls(epilogue_threads):
    """mm2c_mm_chain_dp_batch_host (0: DP + epilogue on the GPU; > 0: epilogue on host threads) against the oracle's
    mm_chain_dp per task; the batch holds empty tasks, tasks without any chain and tasks of several thousand anchors"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _mixed_batch_with_gaps()
    res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=epilogue_threads)
    assert len(res) == off.size - 1
    assert _assert_chains(res, P, 3, 40, off, a, f"epilogue_threads={epilogue_threads}") > 40


def test_batched_mm_chain_dp_pipelined_in_chunks():
    """the same entry with the batch cut into many chunks of whole tasks on two streams (chunk size tuned down): offsets and
    chains of every chunk land behind those of the chunks before it"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _mixed_batch_with_gaps()
    from mm2chain import _native as N
    st0, st1 = N.Stats(), N.Stats()
    mm2chain.tune("pipeline_chunk_anchors", 6000)
    try:
        mm2chain.load().mm2c_get_stats(C.byref(st0))
        res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=0)
        mm2chain.load().mm2c_get_stats(C.byref(st1))
    finally:
        mm2chain.tune("pipeline_chunk_anchors", 20 << 20)
    assert st1.passes - st0.passes > 8, "the batch was not cut into chunks"
    assert _assert_chains(res, P, 3, 40, off, a, "chunked") > 40


def test_concurrent_batch_calls_take_separate_contexts():
    """Three host threads inside mm2c_mm_chain_dp_batch_host at once (a host whose pipeline threads overlap their GPU calls): the library keeps two batch contexts
    (stream set + arenas), the first two callers each take one, the third waits for the first -- every call's chains equal the oracle's, whoever shared what."""
    import threading
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    batches = [_stream("mixed", 120, (500, 3000), seed=300 + k) for k in range(3)]
    out = [None] * 3
    def work(k):
        for _ in range(3):                                   # several rounds, so that the calls really overlap
            out[k] = mm2chain.mm_chain_dp_batch(P, 3, 40, batches[k][0], batches[k][1])
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(3):
        _assert_chains(out[k], P, 3, 40, batches[k][0], batches[k][1], f"concurrent batch call {k}")


@pytest.mark.parametrize("min_cnt,min_sc", [(1, 0), (1, 40), (2, 15), (3, 100), (5, 1000), (0, -5)])
def test_device_epilogue_filter_corners(min_cnt, min_sc, epi_path):
    """chain.c:385-388 with thresholds that keep one-anchor chains (incl. chains that keep only an already taken peak) or
    drop nearly everything"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream("mixed", 12, (50, 2500), seed=91)
    res = mm2chain.mm_chain_dp_batch(P, min_cnt, min_sc, off, a, epilogue_threads=0)
    _assert_chains(res, P, min_cnt, min_sc, off, a, f"min_cnt={min_cnt} min_sc={min_sc}")


def test_device_epilogue_on_dense_and_colinear_streams(epi_path):
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    for profile, seed in (("dense", 5), ("colinear", 6), ("sparse", 7)):
        off, a = _stream(profile, 6, (1000, 5000), seed=seed)
        res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=0)
        _assert_chains(res, P, 3, 40, off, a, profile)


def test_device_epilogue_size_classes_in_one_batch():
    """tasks of every size class of the device epilogue in one batch (LDS classes up to 5120 and 8192 anchors, beyond that the kernels that
    work in HBM), in an order that is not by size; chains against the oracle's mm_chain_dp"""
    import mm2chain
    from mm2chain import params, synth
    P = params.map_ont()
    sizes = [300, 9000, 5120, 6000, 12, 7680, 5121, 20000, 0, 4000, 7681]
    parts, off = [], [0]
    for k, n in enumerate(sizes):
        if n:
            parts.append(synth.make_stream(("mixed", "dense", "colinear")[k % 3], 1, n, seed=300 + k)[1].numpy().view(np.uint64))
        off.append(off[-1] + n)
    off = np.array(off, np.int64); a = np.concatenate(parts)
    res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=0)
    assert _assert_chains(res, P, 3, 40, off, a, "size classes") > 100


def _tandem_task(n_groups, copies, seed):
    """n_groups loci; at each, `copies` chains of 4 anchors that share their reference positions (a tandem repeat in the query:
    same x, query positions max_dist_y apart), so that many chains start at equal x"""
    rng = np.random.default_rng(seed)
    rows = []
    for g in range(n_groups):
        r0 = 100000 + g * 20000
        for c in range(int(copies[g % len(copies)])):
            q0 = 1000 + c * 6000 + int(rng.integers(0, 50))
            for s in range(4):
                rows.append(mk_anchor(0, 1, r0 + 20 * s, q0 + 20 * s))
    return pack(rows)


def test_device_epilogue_equal_first_x_replays_the_reference_sort(epi_path):
    """more than 64 chains with equal first-anchor x: the order is that of radix_sort_128x's passes (ksort.h:101-151), which
    is not stable; also <= 64 chains with ties (insertion sort, stable) and > 64 chains without ties"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    tasks = [_tandem_task(60, [3, 2, 1, 4], 1), _tandem_task(12, [3, 2], 2), _tandem_task(90, [1], 3), _tandem_task(150, [2, 5, 1], 4),
             _tandem_task(300, [2, 3], 5), _tandem_task(1800, [2, 3], 6),      # the last one has more chains than the LDS variant of the replay holds
             _tandem_task(480, [2, 3], 7)]                                     # fits the LDS epilogue (4 800 anchors) with more chains than its counting sort takes
    off = np.concatenate([[0], np.cumsum([t.shape[0] for t in tasks])]).astype(np.int64)
    a = np.concatenate(tasks)
    res = mm2chain.mm_chain_dp_batch(P, 3, 40, off, a, epilogue_threads=0)
    _assert_chains(res, P, 3, 40, off, a, "tandem")
    assert res[0][0].size > 64 and res[3][0].size > 64 and res[4][0].size > 512 and res[5][0].size > 4096 and res[6][0].size > 768 and tasks[6].shape[0] <= 5120
    x_first = [int(res[0][1][i, 0]) for i in np.concatenate([[0], np.cumsum(res[0][0] & np.uint64(0xffffffff))[:-1]]).astype(np.int64)]
    assert len(set(x_first)) < len(x_first), "the test must contain chains that start at equal x"


@pytest.mark.parametrize("seed", range(6))
def test_device_epilogue_on_arbitrary_forests(seed, epi_path):
    """the epilogue kernels on f[] / p[] that no DP produced (random forests: long paths, bushy trees, deep in-chunk chains,
    equal scores) against the ORACLE's backtrack (mm2o_fill_v + mm2o_backtrack = chain.c:106-111,348-422) and, beside it, the library's host
    epilogue; device-resident plan API"""
    import mm2chain
    from mm2chain import params
    rng = np.random.default_rng(1000 + seed)
    P = params.map_ont()
    sizes = [1, 2, 63, 64, 65, 130, 1000, 3000, int(rng.integers(1, 5000)), 0, 4097, 5120, 5121, 7680, 7681] if seed < 2 else \
        [1, 2, 63, 64, 65, 130, 1000, 3000, int(rng.integers(1, 5000)), 0, 4097]   # seeds 0, 1: the boundaries of the LDS size classes too
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    total = int(off[-1])
    a = np.zeros((total, 2), np.uint64)
    f = np.zeros(total, np.int32); p = np.full(total, -1, np.int32)
    for k, n in enumerate(sizes):
        if n == 0:
            continue
        o = int(off[k])
        x = np.sort(rng.integers(0, 50 if seed % 2 else 1 << 30, n).astype(np.uint64)) + (np.uint64(k) << np.uint64(32))
        a[o:o + n, 0] = x
        a[o:o + n, 1] = (np.uint64(15) << np.uint64(32)) | rng.integers(0, 1 << 20, n).astype(np.uint64)
        idx = np.arange(n)
        style = (seed + k) % 4
        if style == 0:      # mostly i-1 (long paths), some roots
            par = np.where(rng.random(n) < 0.97, idx - 1, -1)
        elif style == 1:    # bushy: parent anywhere before i
            par = np.where(rng.random(n) < 0.8, (rng.random(n) * idx).astype(np.int64) - (idx == 0), -1)
        elif style == 2:    # parents a short hop back (deep chains inside a chunk, several children per node)
            par = idx - 1 - rng.integers(0, 4, n)
        else:               # mixture with far parents
            par = np.where(rng.random(n) < 0.5, idx - 1, idx - 1 - rng.integers(0, 200, n))
        if n == 3000 and seed == 3:   # roots only: every anchor is a chain end (more keys than the sorts of the LDS epilogue take in LDS)
            par = np.full(n, -1)
        par = np.where(par < 0, -1, par)
        p[o:o + n] = par
        ff = np.zeros(n, np.int64)
        gain = rng.integers(-30, 25 if seed % 3 else 16, n)
        for i in range(n):
            ff[i] = max(15, (ff[par[i]] if par[i] >= 0 else 0) + 15 + gain[i]) if seed % 3 != 2 else int(rng.integers(0, 60))
        if n == 3000 and seed == 3:
            ff = rng.integers(30, 90, n)          # ... all of them above min_sc
        f[o:o + n] = ff
    ref = [ob.backtrack(2, 30, a[off[k]:off[k + 1]], f[off[k]:off[k + 1]], p[off[k]:off[k + 1]]) for k in range(len(sizes))]
    host = mm2chain.chain_epilogue_host(2, 30, off, a, f, p, n_threads=4)
    for k in range(len(sizes)):
        assert np.array_equal(host[k][0], ref[k][0]) and np.array_equal(host[k][1], ref[k][1]), f"host epilogue, task {k}"
    plan = mm2chain.ChainPlan(P, off)
    d_a = torch.from_numpy(a.view(np.int64)).cuda(); d_f = torch.from_numpy(f).cuda(); d_p = torch.from_numpy(p).cuda()
    u_off, u, b_off, b = plan.chains(d_a, d_f, d_p, 2, 30)
    torch.cuda.synchronize()
    u_off = u_off.cpu().numpy(); b_off = b_off.cpu().numpy()
    u = u.cpu().numpy().view(np.uint64); b = b.cpu().numpy().view(np.uint64)
    assert plan.last_epilogue_ms() > 0
    plan.close()
    n_chains = 0
    for k in range(len(sizes)):
        uk, bk = u[u_off[k]:u_off[k + 1]], b[b_off[k]:b_off[k + 1]]
        assert np.array_equal(uk, ref[k][0]), f"task {k} (n={sizes[k]}): u differs ({uk.size} vs {ref[k][0].size})"
        assert np.array_equal(bk, ref[k][1]), f"task {k} (n={sizes[k]}): b differs"
        n_chains += uk.size
    assert n_chains > 10


def _ref_cl_groups():
    """tests/golden/ref_cl_kernel_fp.npz grouped by scalar set: {(mdx, mdy, bw): (offsets, anchors, f, p)}; every task has one q_span"""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cl_kernel_fp.npz"))
    groups = {}
    for k in range(int(z["n_cases"])):
        key = tuple(int(v) for v in z[f"c{k}_scalars"][:3])
        groups.setdefault(key, []).append(k)
    out = {}
    for key, ks in groups.items():
        off = np.concatenate([[0], np.cumsum([z[f"c{k}_anchors"].shape[0] for k in ks])]).astype(np.int64)
        out[key] = (off, np.concatenate([z[f"c{k}_anchors"] for k in ks]), np.concatenate([z[f"c{k}_f"] for k in ks]), np.concatenate([z[f"c{k}_p"] for k in ks]))
    return z, out


def test_hip_equals_the_references_own_device_kernel(knobs):
    """f[] / p[] the REFERENCE'S OWN device kernel produced (device/minimap2_opencl.cl compiled for the host and called like
    run_chaining_on_hw does; tests/golden/ref_cl_kernel_fp.npz, generator beside it) against the HIP path, element by element:
    through the reference's C++ symbol run_chaining_on_hw (chain_hardware.h:68) and through a device-resident plan with the V2 scalars"""
    import mm2chain
    from mm2chain import params
    z, groups = _ref_cl_groups()
    for k in range(int(z["n_cases"])):
        a, (mdx, mdy, bw, q_span), avg = z[f"c{k}_anchors"], [int(v) for v in z[f"c{k}_scalars"]], float(z[f"c{k}_avg"])
        ns = z[f"c{k}_num_subparts"]
        ret, f, p = mm2chain.run_chaining_on_hw(a.shape[0], mdx, mdy, bw, q_span, avg, a, ns, int(ns.sum()), tid=k)
        assert ret == 0
        assert_same(f, p, z[f"c{k}_f"], z[f"c{k}_p"], None, f"run_chaining_on_hw, case {k} {z[f'c{k}_name']}")
    for (mdx, mdy, bw), (off, a, f_ref, p_ref) in groups.items():
        # V2 scalars: max_skip >= max_iter, the early exit cannot fire -- by default through the hand-written loop with max_skip = max_iter - 1 (the counter cannot
        # pass it), with the knob off through the instantiations without the max-skip machinery (C++ loop)
        for via_loop in (1, 0):
            knobs("noskip_loop", via_loop)
            v = []
            f, p = gpu_batch(params.make_params(mdx, mdy, bw, INT32_MAX, 1024, 1.0, 0, 1, -1, mm2chain.MM2C_F_IGNORE_SEG), off, a, variant=v)
            assert_same(f, p, f_ref, p_ref, off, f"plan with V2 scalars {(mdx, mdy, bw)}, noskip_loop={via_loop}")
            assert ("SKIP=1" in v[0] and "loop=asm" in v[0]) if via_loop else ("SKIP=0" in v[0] and "loop=c++" in v[0]), v
        knobs("noskip_loop", 1)
        # and the stock CPU semantics (V1) restricted to what V2 can express: same vectors
        f, p = gpu_batch(params.make_params(mdx, mdy, bw, max_skip=INT32_MAX, max_iter=1024), off, a)
        assert_same(f, p, f_ref, p_ref, off, f"plan with V1 kernel, max_skip = inf, max_iter = 1024, {(mdx, mdy, bw)}")


@pytest.fixture
def knobs():
    """tuning knobs a test changes, put back afterwards (results never depend on them; the instantiation that runs does)"""
    import mm2chain
    import helpers

    def tune(key, val):
        if key == "coop_plans":
            helpers.PINNED_ROUTE = val                        # (gpu_batch then runs the input once, as pinned, instead of on both routes)
        return mm2chain.tune(key, val)
    yield tune
    helpers.PINNED_ROUTE = None
    for key, val in (("ring_class", int(os.environ.get("MM2C_RING_CLASS", "3"))), ("far_ring", int(os.environ.get("MM2C_FAR_RING", "1"))), ("force_tab", 0), ("noskip_loop", 1), ("compact_ring", 1), ("wide_share_threshold", 40), ("split_streams", 1),
                     ("plan_cut", 1), ("plan_cut_min", 8192), ("seg_min", 256), ("coop_plans", 2), ("coop_waves", 16), ("coop_w8_above", 256), ("fuse_st", 1), ("single_launch", 1), ("coop_max_tasks", 1024), ("q24_ring", 1)):
        mm2chain.tune(key, val)


@pytest.mark.parametrize("route", ["asm", "asm-32-bit-ring", "asm-tab", "asm-tab-32-bit-ring", "asm-short-ring-only", "asm-long-ring-only", "asm-device-cut", "wave-256", "wave-512", "wave-1024",
                                   "coop", "coop-tab", "coop-v2-scalars", "coop8", "coop8-tab", "coop8-v2-scalars", "asm-q24-ring", "asm-tab-q24-ring", "asm-long-32-bit-ring"])
def test_hand_written_loop_equals_the_references_own_device_kernel(route, knobs):
    """The reference-produced vectors through the instantiations that carry the throughput.  With max_skip = 1023 and max_iter = 1024 the
    max-skip machinery of chain.c:226-233 is compiled in and runs (stamps, skip counter, the folds) but cannot fire: among at most 1024
    candidates the counter reaches at most 1023, because the nearest candidate i-1 is never stamped.  The result must therefore be the
    reference kernel's own f[] / p[] (.cl:116-154), and the launcher picks SKIP = true -> the hand-written per-tile loop
    scan_tile_asm_cmp[_far] (look-back 1024 > 448: the `far` instantiation too).  Routes: the gap-cost table (scan_tile_asm_tab[_far]),
    ring-size classes off / forced, tasks cut into pieces on the device, and the first-generation kernel with each of its ring sizes.
    Which instantiation ran is asserted from mm2c_plan_last_variant."""
    from mm2chain import params
    if route in ("asm-tab", "asm-tab-32-bit-ring", "asm-tab-q24-ring"): knobs("force_tab", 1)
    if route.endswith("32-bit-ring") or route.endswith("q24-ring"): knobs("compact_ring", 0)
    if route.endswith("q24-ring") or route == "asm-long-32-bit-ring": knobs("far_ring", 2)      # every task in the long ring: the q24 form (round 5), or its 32-bit form with the knob off
    if route == "asm-long-32-bit-ring": knobs("q24_ring", 0)
    if route == "asm-short-ring-only": knobs("far_ring", 0)
    if route == "asm-long-ring-only": knobs("far_ring", 2)
    if route == "asm-device-cut": knobs("plan_cut_min", 1000); knobs("seg_min", 64)
    if route.startswith("wave-"): knobs("ring_class", {"256": 0, "512": 1, "1024": 2}[route[5:]])
    if route.startswith("coop"): knobs("coop_plans", 1)      # several waves per task (chain_dp_coop.h): what a lone run_chaining_on_hw call runs
    if route.startswith("coop8"): knobs("coop_w8_above", 0)  # ... in the width the launcher takes for more pieces than CUs: eight waves, two workgroups per CU
    if route.endswith("-tab") and route.startswith("coop"): knobs("force_tab", 1)
    # coop-v2-scalars: the scalars of the reference symbol itself (max_skip = INT_MAX, max_iter = 1024), which the launcher maps onto max_skip = max_iter - 1
    ms = INT32_MAX if route.endswith("-v2-scalars") else 1023
    z, groups = _ref_cl_groups()
    assert (5000, 5000, 500) in groups and (10000, 10000, 2000) in groups
    n = 0
    for (mdx, mdy, bw), (off, a, f_ref, p_ref) in groups.items():
        v = []
        f, p = gpu_batch(params.make_params(mdx, mdy, bw, max_skip=ms, max_iter=1024), off, a, variant=v)
        assert_same(f, p, f_ref, p_ref, off, f"{route}, scalars {(mdx, mdy, bw)}: {v[0]}")
        if route.startswith("coop"):
            assert v[0].startswith("chain_dp_coop<W=8," if route.startswith("coop8") else "chain_dp_coop<W=16,") and "FAR=1" in v[0] and "loop=asm" in v[0], v
            assert ("TAB=1" in v[0]) == (route.endswith("-tab") and bw <= 511), v
        elif route.startswith("wave-"):
            assert v[0].startswith(f"chain_dp_wave<R={route[5:]},SKIP=1"), v
        else:
            assert v[0].startswith("chain_dp_tile<") and "SKIP=1" in v[0] and "GEN=0" in v[0] and "FAR=1" in v[0] and "loop=asm" in v[0], v
            assert ("TAB=1" in v[0]) == (route.startswith("asm-tab") and bw <= 511), v
            assert ("compact=1" in v[0]) == (not route.endswith("32-bit-ring") and not route.endswith("q24-ring")), v
            assert ("q24=1" in v[0]) == (route not in ("asm-short-ring-only", "asm-long-32-bit-ring")), v      # wherever there are ring-size classes the long ring is the q24 form
            assert ("classes=1" in v[0]) == (route != "asm-short-ring-only"), v
            assert ("cut=1" in v[0]) == (int(np.diff(off).max()) >= (1000 if route == "asm-device-cut" else 8192)), v   # plan_cut_min
        n += a.shape[0]
    assert n > 150000


def test_hip_on_every_committed_golden_vector():
    """the HIP path over every tests/golden/*.npz regression vector (made by make_golden.py; the CPU suite checks the oracle on them)"""
    from mm2chain import params
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    files = sorted(x for x in os.listdir(g) if x.endswith(".npz") and not x.startswith("ref_"))
    assert len(files) >= 6
    for name in files:
        z = np.load(os.path.join(g, name))
        P = params.make_params(**{k: (float(z["par_" + k]) if k == "gap_scale" else int(z["par_" + k])) for k in
                                  ("max_dist_x", "max_dist_y", "bw", "max_skip", "max_iter", "gap_scale", "is_cdna", "n_segs")})
        f, p = gpu_batch(P, z["offsets"], z["anchors"])
        assert_same(f, p, z["f"], z["p"], z["offsets"], name)


def test_runs_are_ordered_on_torchs_stream_without_device_sync():
    """producer (upload, fill) and consumer (download) on torch's current stream -- the default one (the HIP null stream, handle 0) and a side
    stream -- with no torch.cuda.synchronize() anywhere: the library enqueues on the stream it is handed, NULL meaning the null stream"""
    import contextlib
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream("mixed", 96, (500, 4000), seed=77)
    f_ref, p_ref = oracle_batch(P, off, a)
    total = a.shape[0]
    h_a = torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1, 2)).pin_memory()
    for side in (None, torch.cuda.Stream()):
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            d_a = h_a.cuda(non_blocking=True)
            d_f = torch.empty(total, dtype=torch.int32, device="cuda").fill_(-5)
            d_p = torch.empty(total, dtype=torch.int32, device="cuda").fill_(-5)
            plan = mm2chain.ChainPlan(P, off)
            plan.run(d_a, d_f, d_p)
            u_off, u, b_off, b = plan.chains(d_a, d_f, d_p, 3, 40)
            f, p, n_chains = d_f.cpu().numpy(), d_p.cpu().numpy(), int(u_off[-1].item())     # stream-ordered copies
            plan.close()
        assert_same(f, p, f_ref, p_ref, off, "side stream" if side is not None else "default stream")
        assert n_chains > 0


def test_short_device_buffers_are_refused():
    """the device-pointer entries take the extent of every buffer; one that is shorter than the plan needs is refused with MM2C_E_TOOBIG
    before anything is launched (cf. the reference's n > BUFFER_N check, chain_hardware.cpp:34-37)"""
    import mm2chain
    from mm2chain import params
    P = params.map_ont()
    off, a = _stream("mixed", 8, 1000, seed=3)
    total = a.shape[0]
    d_a = torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1, 2)).cuda()
    d_f = torch.empty(total, dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
    plan = mm2chain.ChainPlan(P, off)
    for bad in ("anchors", "f", "p", "avg"):
        args = dict(anchors=d_a, f=d_f, p=d_p, avg=None)
        if bad == "anchors": args["anchors"] = d_a[:-1].contiguous()
        elif bad == "avg": args["avg"] = torch.zeros(7, dtype=torch.float32, device="cuda")
        else: args[bad] = torch.empty(total - 1, dtype=torch.int32, device="cuda")
        with pytest.raises(RuntimeError, match="shorter than the plan"):
            plan.run(args["anchors"], args["f"], args["p"], args["avg"])
    plan.run(d_a, d_f, d_p)
    with pytest.raises(RuntimeError, match="shorter than the plan"):
        plan.chains(d_a, d_f[:-1].contiguous(), d_p, 3, 40)
    torch.cuda.synchronize()
    plan.close()
    # seed plan: a match that points outside the declared hit pool is caught on the device
    from mm2chain import synth
    m, h = synth.matches_from_anchors(a[off[0]:off[1]], 1 << 20)
    sp = mm2chain.SeedPlan(np.array([0, m.size], np.int64), np.array([0, int(m["n"].sum())], np.int64))
    d_m = torch.from_numpy(m.view(np.uint8).copy()).cuda()
    d_h = torch.from_numpy(h.view(np.int64)).cuda()
    d_q = torch.tensor([1 << 20], dtype=torch.int32, device="cuda")
    sp.run(d_m, d_h, d_q); sp.check()
    sp.run(d_m, d_h[:-1].contiguous(), d_q)
    with pytest.raises(RuntimeError, match="outside the declared hit pool"):
        sp.check()
    sp.close()


def test_real_anchor_lists_from_the_reference_test_data():
    """anchors that reach mm_chain_dp for test/MT-human.fa vs MT-orang.fa and t-inv.fa vs q-inv.fa (dumped through the
    reference's own host objects, tests/golden/make_ref_anchor_fixtures.py): f/p through the kernel, chains through mm_chain_dp"""
    import os
    import mm2chain
    from mm2chain import params
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_testdata_anchors.npz"))
    for k in range(int(z["n_calls"])):
        h = z[f"c{k}_par"]
        a = z[f"c{k}_anchors"]
        P = params.make_params(int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4]), float(h[9]), int(h[7]), int(h[8]))
        f, p = gpu_batch(P, np.array([0, a.shape[0]], dtype=np.int64), a)
        assert_same(f, p, z[f"c{k}_f"], z[f"c{k}_p"], None, f"real anchors call {k}")
        u, b = mm2chain.mm_chain_dp(int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4]), int(h[5]), int(h[6]), float(h[9]),
                                    int(h[7]), int(h[8]), a)
        assert np.array_equal(u, z[f"c{k}_u"]) and np.array_equal(b, z[f"c{k}_b"])
    assert int(z["c0_u"][0] >> np.uint64(32)) == 3189        # PAF s1:i:3189 of the reference


def test_one_long_task_and_many_tiny_tasks():
    """ava-ont-like: one task of 300k anchors (long colinear chains + noise) beside 20k tasks of 1..40 anchors"""
    from mm2chain import params, synth
    P = params.ava_ont()
    _, a_long = synth.make_stream("mixed", 1, 300000, seed=77, locus=3000000)
    sizes = [300000]
    parts = [a_long.numpy().view(np.uint64)]
    rng = np.random.default_rng(5)
    _, a_small = synth.make_stream("dense", 1, 820000, seed=78, locus=40000000)
    a_small = a_small.numpy().view(np.uint64)
    pos = 0
    for _ in range(20000):
        n = int(rng.integers(1, 41))
        assert pos + n <= a_small.shape[0]
        parts.append(a_small[pos:pos + n]); sizes.append(n); pos += n
    a = np.concatenate(parts)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "long + tiny")


def _random_task(rng, n, n_refs, n_segs, dense):
    """adversarial anchors: duplicated x, dq <= 0, huge jumps, several references/strands, random spans and segment ids"""
    rows = []
    for _ in range(n_refs):
        strand, rid = int(rng.integers(0, 2)), int(rng.integers(0, 5))
        pos = int(rng.integers(0, 1 << 20)); q = int(rng.integers(0, 5000))
        for _ in range(n // n_refs):
            r = rng.random()
            pos += 0 if r < .08 else int(rng.integers(1, 12 if dense else 400)) if r < .95 else int(rng.integers(3000, 30000))
            q += int(rng.integers(-40, 60 if dense else 300))
            rows.append(mk_anchor(strand, rid, pos, max(q, 0), span=int(rng.integers(1, 40)), seg=int(rng.integers(0, n_segs))))
    return pack(rows)


@pytest.mark.parametrize("seed", range(12))
def test_randomised_parameters_and_adversarial_anchors(seed, general_kernel):
    """every scalar of mm_chain_dp drawn at random (incl. degenerate values) on adversarial anchor lists"""
    from mm2chain import params
    rng = np.random.default_rng(1000 + seed)
    n_segs = int(rng.choice([1, 1, 2, 3]))
    P = params.make_params(max_dist_x=int(rng.choice([0, 50, 700, 5000, 100000])), max_dist_y=int(rng.choice([-5, 60, 700, 5000])),
                           bw=int(rng.choice([-1, 0, 10, 500, 5000])), max_skip=int(rng.choice([-1, 0, 1, 5, 25, 300, INT32_MAX])),
                           max_iter=int(rng.choice([-3, 0, 1, 63, 64, 65, 200, 5000, INT32_MAX])),
                           gap_scale=float(rng.choice([1.0, 1.0, 0.5, 2.25, 0.0])), is_cdna=int(rng.integers(0, 2)), n_segs=n_segs)
    tasks = [_random_task(rng, int(rng.integers(1, 1500)), int(rng.integers(1, 4)), n_segs, bool(rng.integers(0, 2))) for _ in range(10)]
    a = np.concatenate(tasks)
    off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, f"random seed {seed}: {params.as_dict(P)}")


def test_batched_runner_on_a_stream_file(tmp_path):
    """tools/mm2chain_run: stream file -> mini-batches through the C ABI -> f/p file and chains"""
    import subprocess
    import mm2chain
    from mm2chain import params, stream
    P = params.map_ont()
    off, a = _stream("mixed", 40, (200, 3000), seed=13)
    src, out = tmp_path / "in.mm2a", tmp_path / "out.bin"
    stream.write(src, P, off, a)
    exe = os.path.join(os.path.dirname(mm2chain.LIB_PATH), "tools", "mm2chain_run")
    f_ref, p_ref = oracle_batch(P, off, a)
    want = []
    for k in range(40):
        u, _ = ob.mm_chain_dp(P, 3, 40, a[off[k]:off[k + 1]])
        want += [(k, int(x >> np.uint64(32)), int(x & np.uint64(0xFFFFFFFF))) for x in u]
    for extra in ([], ["-e", "2"]):                     # backtrack on the GPU / on two host threads
        r = subprocess.run([exe, "-b", "20000", "-c", *extra, "-o", str(out), str(src)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        fp = np.fromfile(out, dtype=np.int32)
        assert_same(fp[: f_ref.size], fp[f_ref.size:], f_ref, p_ref, off, "runner")
        got = [tuple(map(int, ln.split("\t")[1:])) for ln in r.stdout.splitlines() if ln.startswith("CH")]
        assert got == want, extra


def test_big_host_batch_is_pipelined_in_chunks():
    """> 2 M anchors from host memory (pageable and page-locked): the two-stream chunked path"""
    import mm2chain
    from mm2chain import params, _native as N
    P = params.map_ont()
    off, a = _stream("mixed", 700, (2000, 5000), seed=71)
    assert a.shape[0] > (1 << 21)
    f_ref, p_ref = oracle_batch(P, off, a)
    mm2chain.tune("pipeline_chunk_anchors", 300000)          # force the chunked two-stream path at this size
    try:
        f, p = mm2chain.chain_batch_host(P, off, a)          # (chunks of about 85 pieces: the default route gives each piece sixteen waves)
        assert_same(f, p, f_ref, p_ref, off, "big pageable batch, pipelined, default route")
        assert "chain_dp_coop" in mm2chain.last_host_variant(), mm2chain.last_host_variant()
        mm2chain.tune("coop_plans", 0); mm2chain.tune("pipe_coop_chunks", 0)
        f, p = mm2chain.chain_batch_host(P, off, a)
    finally:
        mm2chain.tune("pipeline_chunk_anchors", 20 << 20); mm2chain.tune("coop_plans", 2); mm2chain.tune("pipe_coop_chunks", 1)
    assert_same(f, p, f_ref, p_ref, off, "big pageable batch, pipelined")
    if mm2chain.device_count() == 1:                         # (with MM2C_DEVICES naming several slots the batch is split first and its parts are too small for the pipeline)
        assert "compact=1" in mm2chain.last_host_variant() and "loop=asm" in mm2chain.last_host_variant(), mm2chain.last_host_variant()   # the chunks have the prepass classes
    f, p = mm2chain.chain_batch_host(P, off, a)
    assert_same(f, p, f_ref, p_ref, off, "big pageable batch")
    pa = mm2chain.PinnedArray(a.shape, np.uint64); pf = mm2chain.PinnedArray(f.shape, np.int32); pp = mm2chain.PinnedArray(p.shape, np.int32)
    pa.array[:] = a
    mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array)
    assert_same(pf.array, pp.array, f_ref, p_ref, off, "big pinned batch")


def test_full_size_properties():
    """BASELINE config-2 size (5000 anchors per read, many reads): properties that need no oracle at full size, plus a
    sampled oracle check.  f[i] >= span, -1 <= p[i] < i, f[i] - f[p[i]] <= span, replicated tasks give replicated output."""
    from mm2chain import params, synth
    P = params.map_ont()
    off1, a1 = synth.make_stream("mixed", 512, 5000, seed=2, device="cuda")
    off, a = synth.replicate(off1, a1, 8)
    import mm2chain
    total = a.shape[0]
    d_f = torch.empty(total, dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
    plan = mm2chain.ChainPlan(P, off.numpy())
    plan.run(a, d_f, d_p)
    torch.cuda.synchronize()
    f = d_f.view(8, -1); p = d_p.view(8, -1)
    assert bool((f == f[0]).all()) and bool((p == p[0]).all()), "replicas differ"
    f0 = d_f[: total // 8].cpu().numpy(); p0 = d_p[: total // 8].cpu().numpy()
    idx = np.arange(f0.size) % 5000
    assert (f0 >= 15).all() and (p0 >= -1).all() and (p0 < idx).all()
    has = p0 >= 0
    base = (np.arange(f0.size) // 5000) * 5000
    assert (f0[has] - f0[(base + p0)[has]] <= 15).all()
    a_np = a1.cpu().numpy().view(np.uint64); off_np = off1.numpy()
    sel = slice(0, int(off_np[16]))
    f_ref, p_ref = oracle_batch(P, off_np[:17], a_np[sel])
    assert_same(f0[sel], p0[sel], f_ref, p_ref, off_np[:17], "full-size sample")


def _timed(fn):
    import time
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3
