"""CPU tests (-m "not gpu"): the oracle against hand-derived known answers, an independent pure-Python restatement,
the literal FPGA-kernel emulation, the wave-parallel formulation model, and the committed golden fixtures."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from helpers import mk_anchor, pack
from wave_model import chain_wave_model

INT32_MAX = 2**31 - 1
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class P:  # plain parameter holder with the oracle's field names
    def __init__(self, **kw):
        d = dict(max_dist_x=5000, max_dist_y=5000, bw=500, max_skip=25, max_iter=5000, gap_scale=1.0, is_cdna=0, n_segs=1)
        d.update(kw)
        self.__dict__.update(d)


def py_chain(par, a, avg):
    """independent literal restatement in Python, written from the recurrence in tex/minimap2.tex:105-150 and
    chain.c:184-238 (small inputs only)"""
    import math
    n = a.shape[0]
    f = [0] * n; p = [-1] * n; t = [0] * n
    st = 0
    f32 = np.float32
    for i in range(n):
        xi = int(a[i, 0]); yi = int(a[i, 1])
        qi = (yi & 0xFFFFFFFF); qi = qi - (1 << 32) if qi >= (1 << 31) else qi
        span = (yi >> 32) & 0xff; segi = (yi >> 48) & 0xff
        while st < i and xi > int(a[st, 0]) + par.max_dist_x:
            st += 1
        if i - st > par.max_iter:
            st = i - par.max_iter
        best, bj, nskip = span, -1, 0
        for j in range(i - 1, st - 1, -1):
            xj = int(a[j, 0]); yj = int(a[j, 1])
            qj = (yj & 0xFFFFFFFF); qj = qj - (1 << 32) if qj >= (1 << 31) else qj
            segj = (yj >> 48) & 0xff
            dr = xi - xj; dq = qi - qj; same = segi == segj
            if (same and dr == 0) or dq <= 0: continue
            if (same and dq > par.max_dist_y) or dq > par.max_dist_x: continue
            dd = abs(dr - dq)
            if same and dd > par.bw: continue
            if par.n_segs > 1 and not par.is_cdna and same and dr > par.max_dist_y: continue
            sc = min(dq, dr, span)
            lg = dd.bit_length() - 1 if dd else 0
            lin = int(f32(dd) * f32(avg))
            if par.is_cdna or not same:
                if not same and dr == 0: sc += 1; gap = 0
                elif dr > dq or not same: gap = min(lin, lg)
                else: gap = lin + (lg >> 1)
            else:
                gap = lin + (lg >> 1)
            sc -= int(float(gap) * float(f32(par.gap_scale)) + .499)
            sc += f[j]
            if sc > best:
                best, bj = sc, j
                if nskip > 0: nskip -= 1
            elif t[j] == i + 1:
                nskip += 1
                if nskip > par.max_skip: break
            if p[j] >= 0: t[p[j]] = i + 1
        f[i], p[i] = best, bj
    return np.array(f, np.int32), np.array(p, np.int32)


def test_known_answers_by_hand():
    """three colinear anchors, span 15, avg = .15: worked by hand from chain.c:207-220.
    a0 (r=1000,q=100): f=15.  a1 (r=1010,q=110): dr=dq=10, dd=0 -> sc=10+15=25, p=0.
    a2 (r=1040,q=138): vs a1 dr=30,dq=28,dd=2: min(28,30,15)=15, cost=(int)(2*.15)+(1>>1)=0 -> 15+25=40;
                       vs a0 dr=40,dq=38,dd=2 -> 15+15=30.  f=40,p=1."""
    a = pack([mk_anchor(0, 0, 1000, 100), mk_anchor(0, 0, 1010, 110), mk_anchor(0, 0, 1040, 138)])
    assert abs(ob.avg_qspan(a) - 0.15) < 1e-7
    f, p, v = ob.chain_fpv(P(), a)
    assert f.tolist() == [15, 25, 40] and p.tolist() == [-1, 0, 1] and v.tolist() == [15, 25, 40]
    # a gap: a3 (r=1300,q=300): vs a2 dr=260,dq=162,dd=98: sc=15, cost=(int)(98*.15)=14 + (6>>1)=3 -> 15-17+40=38, p=2
    a = pack([mk_anchor(0, 0, 1000, 100), mk_anchor(0, 0, 1010, 110), mk_anchor(0, 0, 1040, 138), mk_anchor(0, 0, 1300, 300)])
    f, p, v = ob.chain_fpv(P(), a)
    assert f.tolist() == [15, 25, 40, 38] and p.tolist() == [-1, 0, 1, 2] and v.tolist() == [15, 25, 40, 40]


def test_hazards_by_hand():
    # dq <= 0 and dr == 0 are skipped (chain.c:202); dd > bw skipped (:205); other strand / reference never chains
    a = pack([mk_anchor(0, 0, 1000, 100), mk_anchor(0, 0, 1000, 120), mk_anchor(0, 0, 1020, 100),
              mk_anchor(0, 0, 2000, 400), mk_anchor(1, 0, 1010, 110), mk_anchor(0, 1, 1010, 110)])
    f, p, _ = ob.chain_fpv(P(), a)
    # sorted order: (0,0,1000,100) (0,0,1000,120) (0,0,1020,100) (0,0,2000,400) (0,1,1010,110) (1,0,1010,110)
    assert p[1] == -1 and p[2] == -1 and p[4] == -1 and p[5] == -1
    assert p[3] == -1 and f[3] == 15          # dr=980..1000, dq=280..300: dd=700 > bw
    # tie: two predecessors with the same score -> the nearer (larger j) wins (chain.c:226 strict >)
    a = pack([mk_anchor(0, 0, 1000, 100), mk_anchor(0, 0, 1001, 101), mk_anchor(0, 0, 1100, 200)])
    f, p, _ = ob.chain_fpv(P(max_skip=INT32_MAX), a)
    assert f[1] == 16 and p[1] == 0
    # candidates for a2: via a1 -> 15 - cost(dd=0) + 16 = 31 ; via a0 -> 15 + 15 = 30
    assert f[2] == 31 and p[2] == 1
    # sc == q_span must not set p (strict >): one predecessor whose extension scores exactly span
    a = pack([mk_anchor(0, 0, 1000, 100), mk_anchor(0, 0, 1400, 420)])   # dd=80: cost=(int)(80*.15)=12 + (6>>1)=3 = 15 -> 15-15+15 = 15
    f, p, _ = ob.chain_fpv(P(), a)
    assert f.tolist() == [15, 15] and p.tolist() == [-1, -1]


@pytest.mark.parametrize("seed", range(6))
def test_oracle_equals_python_restatement(seed):
    rng = np.random.default_rng(seed)
    par = P(max_skip=int(rng.choice([0, 2, 25, INT32_MAX])), max_iter=int(rng.choice([20, 100, 5000])),
            gap_scale=float(rng.choice([1.0, 0.8, 1.3])), bw=int(rng.choice([100, 500])),
            is_cdna=int(rng.integers(0, 2)), n_segs=int(rng.integers(1, 3)), max_dist_x=int(rng.choice([300, 5000])),
            max_dist_y=int(rng.choice([300, 5000])))
    rows, pos, q = [], 5000, 200
    for _ in range(400):
        pos += int(rng.integers(0, 25)); q += int(rng.integers(-15, 35))
        rows.append(mk_anchor(int(rng.integers(0, 2)) if rng.random() < .02 else 0, 1, pos, max(q, 1),
                              span=int(rng.integers(10, 25)), seg=int(rng.integers(0, par.n_segs))))
    a = pack(rows)
    avg = ob.avg_qspan(a)
    f, p, _ = ob.chain_fpv(par, a, avg)
    f2, p2 = py_chain(par, a, avg)
    assert np.array_equal(f, f2) and np.array_equal(p, p2)


def _stream(profile, n_reads, n_per, seed, **kw):
    from mm2chain import synth
    off, a = synth.make_stream(profile, n_reads, n_per, seed=seed, **kw)
    return off.numpy(), a.numpy().view(np.uint64)


@pytest.mark.parametrize("profile", ["sparse", "mixed", "dense", "colinear"])
def test_wave_formulation_model_equals_oracle(profile):
    """the chunked / prefix-scan formulation the HIP kernel implements, incl. ring aliasing and the far path"""
    off, a = _stream(profile, 2, 900, seed=4)
    for par, R in ((P(), 128), (P(max_skip=2, max_iter=300), 256), (P(max_skip=INT32_MAX, max_iter=150), 128)):
        for k in range(2):
            t = a[off[k]:off[k + 1]]
            avg = ob.avg_qspan(t)
            f, p, _ = ob.chain_fpv(par, t, avg)
            fm, pm = chain_wave_model(par, t, avg, R=R)
            assert np.array_equal(f, fm) and np.array_equal(p, pm)


@pytest.mark.parametrize("profile", ["mixed", "dense", "colinear"])
def test_tile_formulation_model_equals_oracle(profile):
    """the tile-aligned formulation of the round-2 kernel (csrc/chain_dp_tile.h): rings addressed by the anchor index, three-instruction
    filter, equal-x runs, 16-bit / far stamps, fold paths A / B1 / B2"""
    from tile_model import chain_tile_model
    off, a = _stream(profile, 2, 900, seed=4)
    for par in (P(), P(max_skip=2, max_iter=300), P(max_skip=INT32_MAX, max_iter=150), P(gap_scale=0.8, bw=100),
                P(max_iter=5000, max_dist_x=20000, max_dist_y=20000), P(max_skip=-1), P(max_skip=0), P(bw=-1), P(bw=0)):
        for k in range(2):
            t = a[off[k]:off[k + 1]]
            avg = ob.avg_qspan(t)
            f, p, _ = ob.chain_fpv(par, t, avg)
            st = {}
            fm, pm = chain_tile_model(par, t, avg, stats=st)
            assert np.array_equal(f, fm) and np.array_equal(p, pm)
            assert st["anchors"] == t.shape[0] and st["fold_a"] + st["fold_b0"] + st["fold_b1"] + st["fold_b2_closed"] + st["fold_b2_scan"] == \
                st["own_pass"] + st["ring_pass"] + st["far_pass"]


def test_model_of_the_hand_written_loop_reaches_every_label():
    """The NumPy model of the tile kernel's control flow (tests/tile_model.py), in the 32-bit and in the compact ring form, equals the oracle on the inputs
    written for the rare exits of the fold, and those inputs drive every sub-path the assembly has a label for (chain.c:226-233: new maximum, skip event, the
    `break` in fold A, in the closed forms and inside the max-plus scan; deep f / p, partly covered tiles, look-back and stamps beyond the ring) -- a label that
    goes cold fails here, on the CPU.  The real instruction sequence is counted on the GPU with the same inputs among the others (tests/test_gpu_labels.py,
    profiles/r4_label_hits.md)."""
    from helpers import fold_driver_tasks
    from tile_model import chain_tile_model, LABEL_KEYS
    rng = np.random.default_rng(4242)
    tasks = fold_driver_tasks(rng, shapes=((1100, 4, 0.05, 0.15), (2000, 3, 0.0, 0.3), (700, 12, 0.2, 0.15), (500, 40, 0.0, 0.2)))
    off, a = _stream("mixed", 2, 700, seed=5)                 # noise around a chain: scans that run to the end of their window (partly covered tiles, lone candidates)
    tasks += [a[off[0]:off[1]], a[off[1]:off[2]]]
    off, a = _stream("dense", 1, 1600, seed=8, locus=5000)    # noise in one window of 1 600 anchors: scans that go on beyond a ring of 16 tiles
    tasks.append(a)
    tot = {}
    for form, (compact, NX) in {"32-bit ring of 8 tiles": (False, 8), "compact ring of 16 tiles": (True, 16)}.items():
        S = dict.fromkeys(LABEL_KEYS, 0)
        for max_skip in (1, 3, 25):
            par = P(max_skip=max_skip)
            for t in tasks:
                avg = ob.avg_qspan(t)
                f, p, _ = ob.chain_fpv(par, t, avg)
                st = {}
                fm, pm = chain_tile_model(par, t, avg, NX=NX, stats=st, compact=compact)
                assert np.array_equal(f, fm) and np.array_equal(p, pm), (form, max_skip)
                for k in LABEL_KEYS:
                    S[k] += st[k]
        tot[form] = S
        cold = [k for k in LABEL_KEYS if S[k] < 3]
        assert not cold, f"{form}: sub-paths of the fold the driver inputs no longer reach: {cold}"
    assert tot["32-bit ring of 8 tiles"]["far_pass"] > 100 and tot["compact ring of 16 tiles"]["far_pass"] > 100


def test_fpga_literal_equals_v1_with_v2_parameters():
    """device/minimap2_opencl.cl emulated literally == chain.c loop with max_skip=inf, max_iter=1024"""
    off, a = _stream("dense", 2, 2500, seed=8, locus=9000)
    for k in range(2):
        t = a[off[k]:off[k + 1]]
        avg = ob.avg_qspan(t)
        f_lit, p_lit = ob.chain_hw_literal(5000, 5000, 500, 15, avg, t)
        f, p, _ = ob.chain_fpv(P(max_skip=INT32_MAX, max_iter=1024), t, avg)
        assert np.array_equal(f, f_lit) and np.array_equal(p, p_lit)
        ns, tot, trip = ob.predict(t, 5000)
        assert tot == int(ns.sum()) and ns.min() >= 1 and ns.max() <= 8 and trip <= 1024 * t.shape[0]


def test_radix_sort_restatement():
    import ctypes as C
    rng = np.random.default_rng(0)
    lib = ob.load()
    for n in (0, 1, 5, 64, 65, 1000, 70000):
        u = rng.integers(0, 2**63, n, dtype=np.uint64)
        v = u.copy()
        lib.mm2o_radix_sort_64(v.ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(v, np.sort(u))
        w = np.stack((rng.integers(0, 50, n).astype(np.uint64) << np.uint64(20), np.arange(n, dtype=np.uint64)), 1)
        w = np.ascontiguousarray(w); w0 = w.copy()
        lib.mm2o_radix_sort_128x(w.ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(w[:, 0], np.sort(w0[:, 0])) and np.array_equal(np.sort(w[:, 1]), np.arange(n, dtype=np.uint64))
        if n <= 64:   # insertion sort is stable
            assert np.array_equal(w, w0[np.argsort(w0[:, 0], kind="stable")])


def test_mm_chain_dp_oracle_properties():
    off, a = _stream("mixed", 3, 2500, seed=12)
    for k in range(3):
        t = a[off[k]:off[k + 1]]
        u, b = ob.mm_chain_dp(P(), 3, 40, t)
        cnt = (u & np.uint64(0xFFFFFFFF)).astype(np.int64)
        assert b.shape[0] == cnt.sum() and (cnt >= 3).all() and ((u >> np.uint64(32)) >= 40).all()
        # chains are emitted in ascending x of their first anchor, anchors inside a chain ascending in x and q
        starts = np.concatenate(([0], np.cumsum(cnt)[:-1]))
        assert (np.diff(b[starts, 0].astype(object)) >= 0).all()
        for s, c in zip(starts, cnt):
            seg = b[s:s + c]
            assert (np.diff(seg[:, 0].astype(object)) >= 0).all()
            assert (np.diff((seg[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)) > 0).all()
        # every chained anchor is an input anchor
        keys = set(map(tuple, t.tolist()))
        assert all(tuple(r) in keys for r in b.tolist())


def test_golden_fixtures():
    """committed vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle; see the README there)"""
    files = sorted(x for x in os.listdir(GOLDEN) if x.endswith(".npz") and not x.startswith("ref_"))
    assert files, "no golden fixtures"
    for name in files:
        z = np.load(os.path.join(GOLDEN, name))
        par = P(**{k: (float(z["par_" + k]) if k == "gap_scale" else int(z["par_" + k])) for k in
                   ("max_dist_x", "max_dist_y", "bw", "max_skip", "max_iter", "gap_scale", "is_cdna", "n_segs")})
        off, a = z["offsets"], z["anchors"]
        f, p, _ = ob.chain_batch(par, off, a, 2)
        assert np.array_equal(f, z["f"]) and np.array_equal(p, z["p"]), name


def test_oracle_equals_the_references_own_device_kernel():
    """tests/golden/ref_cl_kernel_fp.npz holds f[] / p[] PRODUCED BY THE REFERENCE: device/minimap2_opencl.cl compiled for the host by
    the image's clang and called as run_chaining_on_hw calls it (tests/golden/make_ref_cl_fixtures.py, oracle/ref_host/Makefile).
    Both restatements of the oracle must reproduce them element by element: the literal emulation of the kernel's control flow
    (mm2o_chain_hw_literal) and the stock CPU loop chain.c:184-238 (mm2o_chain_fpv) run with the V2 scalars max_skip = inf,
    max_iter = 1024 -- which pins the window, the filters chain.c:202-205, the score chain.c:207-220 incl. the f32
    truncation and ilog2, strict-max / nearest-j tie rule and the p = -1 rule of the CPU loop to reference-produced values."""
    z = np.load(os.path.join(GOLDEN, "ref_cl_kernel_fp.npz"))
    n_cases, n_anch, n_deep = int(z["n_cases"]), 0, 0
    assert n_cases >= 18
    for k in range(n_cases):
        a, (mdx, mdy, bw, q_span), avg = z[f"c{k}_anchors"], [int(v) for v in z[f"c{k}_scalars"]], float(z[f"c{k}_avg"])
        ns, tot, _ = ob.predict(a, mdx)
        assert np.array_equal(ns, z[f"c{k}_num_subparts"])
        f_lit, p_lit = ob.chain_hw_literal(mdx, mdy, bw, q_span, avg, a)
        assert np.array_equal(f_lit, z[f"c{k}_f"]) and np.array_equal(p_lit, z[f"c{k}_p"]), f"literal emulation, case {k} {z[f'c{k}_name']}"
        par = P(max_dist_x=mdx, max_dist_y=mdy, bw=bw, max_skip=2**31 - 1, max_iter=1024)
        f, p, _ = ob.chain_fpv(par, a, avg)
        assert np.array_equal(f, z[f"c{k}_f"]) and np.array_equal(p, z[f"c{k}_p"]), f"chain.c loop with V2 scalars, case {k} {z[f'c{k}_name']}"
        # the same loop with the max-skip machinery of chain.c:226-233 switched ON but unable to fire: inside 1024 candidates the counter
        # reaches at most 1023 (the nearest candidate i-1 is never stamped), so max_skip = 1023 must give the reference's values too.
        # The GPU test runs these scalars through the hand-written loop (SKIP = true instantiation); this pins the oracle it is compared with.
        par = P(max_dist_x=mdx, max_dist_y=mdy, bw=bw, max_skip=1023, max_iter=1024)
        f, p, _ = ob.chain_fpv(par, a, avg)
        assert np.array_equal(f, z[f"c{k}_f"]) and np.array_equal(p, z[f"c{k}_p"]), f"chain.c loop, max_skip 1023 / max_iter 1024, case {k} {z[f'c{k}_name']}"
        n_anch += a.shape[0]; n_deep += int((ns == 8).sum())
    assert n_anch > 150000 and n_deep > 3000     # anchors whose window fills all 8 sub-parts (look-back of 1024)
    assert n_cases >= 35                         # incl. the first reads of bench.py's own streams under map-ont / asm20 / ava-ont scalars


def test_host_epilogue_of_the_library_equals_the_oracle_mm_chain_dp():
    """mm2c_chain_epilogue_host (library, host threads, no GPU) on the oracle's f[] / p[] against the oracle's mm_chain_dp"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minimap2-fpga_amd"))
    import mm2chain
    from mm2chain import params, synth
    P = params.map_ont()
    for profile, seed in (("mixed", 3), ("dense", 4), ("sparse", 5)):
        off, a = synth.make_stream(profile, 12, (1, 3000), seed=seed)
        off = off.numpy(); a = a.numpy().view(np.uint64)
        f, p, _ = ob.chain_batch(P, off, a, n_threads=4)
        for min_cnt, min_sc in ((3, 40), (1, 0)):
            res = mm2chain.chain_epilogue_host(min_cnt, min_sc, off, a, f, p, n_threads=3)
            for k in range(off.size - 1):
                u_ref, b_ref = ob.mm_chain_dp(P, min_cnt, min_sc, a[off[k]:off[k + 1]])
                assert np.array_equal(res[k][0], u_ref) and np.array_equal(res[k][1], b_ref), (profile, min_cnt, k)


def test_seed_hits_oracle_equals_the_reference_anchor_lists():
    """mm2o_collect_seed_hits (map.c:215-247 restated, incl. the unstable radix_sort_128x) on the matches of the committed fixture
    against the anchor lists the reference's own map.o produced for the same reads (tests/golden/make_ref_seed_fixtures.py);
    four of the reads contain anchors with equal x, where the order is that sort's"""
    d = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz"))
    n_ties = 0
    for k in range(int(d["n_reads"])):
        a = ob.collect_seed_hits(d[f"r{k}_matches"], d[f"r{k}_hits"], int(d[f"r{k}_qlen"]))
        ref = d[f"r{k}_anchors"]
        assert np.array_equal(a, ref), f"read {k} ({d[f'r{k}_src']}): anchors differ from the reference's"
        n_ties += int((ref[1:, 0] == ref[:-1, 0]).sum())
    assert n_ties > 1000


def test_seed_hits_oracle_with_skip_seed_equals_the_reference_all_vs_all_anchor_lists():
    """`-x ava-ont` (options.c:82-86: NO_DIAG | NO_DUAL): mm2o_collect_seed_hits_flags -- skip_seed (map.c:122-147) with the name comparison
    carried by ranks, MM_SEED_SELF (map.c:241) -- against the anchor lists the reference's own map.o handed to mm_chain_dp when 37 reads
    were mapped against themselves (tests/golden/make_ref_ava_fixtures.py); incl. a read whose name matches another of different length"""
    d = np.load(os.path.join(GOLDEN, "ref_seed_hits_ava.npz"))
    n_self = n_drop = 0
    for k in range(int(d["n_reads"])):
        a = ob.collect_seed_hits(d[f"r{k}_matches"], d[f"r{k}_hits"], int(d[f"r{k}_qlen"]), int(d["flag"]), d["ref_rank"], d["ref_len"],
                                 int(d[f"r{k}_qlo"]), int(d[f"r{k}_qeq"]))
        ref = d[f"r{k}_anchors"]
        assert np.array_equal(a, ref), f"read {k}: anchors differ from the reference's ({a.shape[0]} vs {ref.shape[0]})"
        n_self += int(((ref[:, 1] >> np.uint64(43)) & np.uint64(1)).sum())
        n_drop += int(d[f"r{k}_matches"]["n"].sum()) - ref.shape[0]
        # without the flags every hit becomes an anchor: the fixture really exercises the skipping
        assert ob.collect_seed_hits(d[f"r{k}_matches"], d[f"r{k}_hits"], int(d[f"r{k}_qlen"])).shape[0] == int(d[f"r{k}_matches"]["n"].sum())
    assert n_self > 20 and n_drop > 30000


def test_seed_hits_heap_oracle_equals_the_reference_anchor_lists():
    """mm2o_collect_seed_hits_heap (collect_seed_hits_heap, map.c:149-213, with the binary heap of ksort.h:43-60 restated operation by operation)
    against the anchor lists the reference's own map.o produced for the reads of ref_seed_hits.npz under MM_F_HEAP_SORT (--heap-sort=yes;
    tests/golden/make_ref_heap_fixtures.py): the same anchors as the radix-sorted lists, in four reads at other places among equal x; and under
    ava-ont flags the heap form keeps exactly the anchors the radix form keeps"""
    d, h = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz")), np.load(os.path.join(GOLDEN, "ref_seed_hits_heap.npz"))
    assert int(h["n_reads"]) == int(d["n_reads"])
    n_moved = 0
    for k in range(int(d["n_reads"])):
        a = ob.collect_seed_hits(d[f"r{k}_matches"], d[f"r{k}_hits"], int(d[f"r{k}_qlen"]), heap=True)
        ref = h[f"r{k}_anchors_heap"]
        assert np.array_equal(a, ref), f"read {k}: heap-merged anchors differ from the reference's"
        n_moved += int((ref != d[f"r{k}_anchors"]).any(axis=1).sum())
    assert n_moved > 2000
    v = np.load(os.path.join(GOLDEN, "ref_seed_hits_ava.npz"))
    for k in range(int(v["n_reads"])):
        args = (v[f"r{k}_matches"], v[f"r{k}_hits"], int(v[f"r{k}_qlen"]), int(v["flag"]), v["ref_rank"], v["ref_len"], int(v[f"r{k}_qlo"]), int(v[f"r{k}_qeq"]))
        a, b = ob.collect_seed_hits(*args, heap=True), ob.collect_seed_hits(*args)
        key = lambda t: np.sort(np.ascontiguousarray(t).view([("x", "<u8"), ("y", "<u8")]).ravel(), order=("x", "y"))
        assert a.shape == b.shape and np.array_equal(key(a), key(b)) and np.array_equal(a[:, 0], b[:, 0]), k


def test_matches_from_anchors_expand_back_into_the_same_anchors():
    """synth.matches_from_anchors (the seed-hit input of bench.py and tools/seed_probe.py): collect_seed_hits of the derived matches gives the
    anchors of the stream again (as a multiset, sorted by x; the order among equal x is the sort's)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minimap2-fpga_amd"))
    from mm2chain import synth
    for profile in ("mixed", "dense", "sparse", "colinear"):
        off, a = synth.make_stream(profile, 3, (1, 1500), seed=17)
        off = off.numpy(); a = a.numpy().view(np.uint64)
        for k in range(3):
            t = a[off[k]:off[k + 1]]
            m, h = synth.matches_from_anchors(t, 1 << 20)
            assert int(m["n"].sum()) == t.shape[0] and h.size == t.shape[0]
            got = ob.collect_seed_hits(m, h, 1 << 20)
            assert np.all(got[1:, 0] >= got[:-1, 0])
            key = lambda z: z[np.lexsort((z[:, 1], z[:, 0]))]
            assert np.array_equal(key(got), key(t)), (profile, k)


def test_compact_ring_arithmetic_is_exact_within_its_bound():
    """The tile kernel's compact x / q ring keeps the low 16 bits of x and q (chain_dp_tile.h, Lds<>): inside a window (0 <= x_i - x_j <= max_dist_x < 2^16)
    the differences of the low halves mod 2^16 must give the filter of chain.c:202-205 the same verdict, and the same dr / dq for the pairs that pass, as the
    full-width arithmetic -- for every task whose q values span at most 65535 - max_dq; one more and a pair can alias (the bound is tight).  NumPy restatement
    of both sides on random windows, no GPU."""
    rng = np.random.default_rng(5)
    for max_dist_x, max_dist_y, bw in ((5000, 5000, 500), (10000, 10000, 2000), (65535, 20000, 3000), (700, 60, 10)):
        max_dq = min(max_dist_x, max_dist_y)
        bound = 65535 - max_dq
        for trial in range(40):
            n = 400
            x = np.sort(rng.integers(0, 3 * max_dist_x + 2, n)) + int(rng.integers(0, 1 << 31)) + (int(rng.integers(0, 1 << 30)) << 32)   # rid / strand bits above
            span = bound if trial % 2 else int(rng.integers(0, bound + 1))
            q = int(rng.integers(-(1 << 31), (1 << 31) - span)) + rng.integers(0, span + 1, n)
            q[0], q[1] = q.min(), q.min() + span                     # the span is really reached
            i, j = np.triu_indices(n, 1)
            i, j = j, i                                              # j < i
            in_w = x[i] - x[j] <= max_dist_x                         # the window of chain.c:192
            i, j = i[in_w], j[in_w]
            dr, dq = x[i] - x[j], q[i] - q[j]
            ok = (dr > 0) & (dq > 0) & (dq <= max_dq) & (np.abs(dr - dq) <= bw)
            dr16 = ((x[i] & 0xffff) - 1 - (x[j] & 0xffff)) & 0xffff  # dr - 1 and dq - 1 as the 16-bit subtractions leave them, zero-extended
            dq16 = ((q[i] & 0xffff) - 1 - (q[j] & 0xffff)) & 0xffff
            dd16 = np.abs(dr16 - dq16)
            ok16 = (dr16 != 0xffff) & (dq16 <= max_dq - 1) & (dd16 <= bw)   # dr == 0 never reaches the filter (equal-x runs are masked): tested here as -1
            assert np.array_equal(ok, ok16), (max_dist_x, trial)
            assert np.array_equal(dr16[ok], dr[ok] - 1) and np.array_equal(dq16[ok], dq[ok] - 1)
        # one beyond the bound: a pair with dq = -(bound + 1) passes the 16-bit test as dq = max_dq
        qi, qj = 0, bound + 1
        assert ((qi & 0xffff) - 1 - (qj & 0xffff)) & 0xffff == max_dq - 1


def test_walk_restatement_of_the_tie_replay_equals_the_cycle_leader_loop():
    """csrc/radix_replay.h replays a pass of radix_sort_128x (ksort.h:117-131) as a walk that reads digits only and records source -> destination.
    The restatement, and the control flow of the hand-written loop built on it, against the literal loop on random digit arrays: few and many buckets,
    skewed digits (long in-place runs at the head), buckets that are exhausted before they become the head."""
    import random
    import replay_model as rm
    rnd = random.Random(12345)
    for trial in range(6000):
        K = rnd.choice([2, 3, 4, 7, 16, 256])
        n = rnd.randint(1, 300)
        if rnd.random() < 0.3:
            dig = [rnd.randrange(K) if rnd.random() < 0.5 else 0 for _ in range(n)]
        elif rnd.random() < 0.2:
            dig = sorted(rnd.randrange(K) for _ in range(n))          # already distributed: every record in place
        else:
            dig = [rnd.randrange(K) for _ in range(n)]
        ids = rm.literal(dig, K)
        assert rm.walk(dig, K) == ids, (trial, dig)
        assert rm.asm_walk(dig, K) == ids, (trial, dig)
