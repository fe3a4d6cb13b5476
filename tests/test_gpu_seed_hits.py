"""GPU parity tests of the seed-hit path: matches in, the anchor array collect_seed_hits (map.c:215-247)
hands to mm_chain_dp out -- against anchor lists produced by the reference's own map.o (committed fixture) and against the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle_binding as ob

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def _init():
    import mm2chain
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    mm2chain.init()
    yield
    mm2chain.shutdown()


def _batch(reads):
    """reads: list of (qlen, matches with read-local cr_off, hits) -> one CSR batch with a shared hit pool"""
    mo, ms, hs, ql, base = [0], [], [], [], 0
    for qlen, m, hits in reads:
        m = np.array(m, dtype=ob.MATCH_DTYPE, copy=True)
        m["cr_off"] += base
        base += hits.size
        ms.append(m); hs.append(np.asarray(hits, np.uint64)); ql.append(qlen); mo.append(mo[-1] + m.size)
    return (np.array(mo, np.int64), np.concatenate(ms) if ms else np.zeros(0, ob.MATCH_DTYPE),
            np.concatenate(hs) if hs else np.zeros(0, np.uint64), np.array(ql, np.int32))


def _random_read(rng, n_matches, max_n, rid_count, pos_range, qlen=12000, dup_frac=0.0):
    """matches with random hit lists; a small pos_range or dup_frac > 0 (the same hit list under two query minimizers, as a repeat in
    the query gives) makes anchors with equal x"""
    m = np.zeros(n_matches, ob.MATCH_DTYPE)
    m["n"] = rng.integers(0, max_n + 1, n_matches)
    m["q_pos"] = (np.sort(rng.integers(15, qlen, n_matches)).astype(np.uint32) << 1) | rng.integers(0, 2, n_matches).astype(np.uint32)
    m["q_span"] = 15
    m["seg_tandem"] = rng.integers(0, 2, n_matches)
    lists = []
    for k in range(n_matches):
        n = int(m["n"][k])
        if k > 0 and dup_frac > 0 and rng.random() < dup_frac and lists[-1].size == n:
            lists.append(lists[-1].copy())
            continue
        rid = rng.integers(0, rid_count, n).astype(np.uint64)
        pos = np.sort(rng.integers(0, pos_range, n)).astype(np.uint64)
        lists.append((rid << np.uint64(32)) | (pos << np.uint64(1)) | rng.integers(0, 2, n).astype(np.uint64))
    m["n"] = [x.size for x in lists]
    m["cr_off"] = np.concatenate([[0], np.cumsum(m["n"].astype(np.int64))[:-1]])
    return qlen, m, (np.concatenate(lists) if lists else np.zeros(0, np.uint64))


def _check(reads, what):
    import mm2chain
    mo, m, h, ql = _batch(reads)
    ao, a = mm2chain.seed_hits_batch(mo, m, h, ql)
    n_ties = 0
    for r, (qlen, mr, hr) in enumerate(reads):
        ref = ob.collect_seed_hits(mr, hr, qlen)
        got = a[ao[r]:ao[r + 1]]
        assert got.shape == ref.shape, f"{what}: read {r}: {got.shape[0]} anchors, expected {ref.shape[0]}"
        bad = np.nonzero((got != ref).any(axis=1))[0]
        assert bad.size == 0, f"{what}: read {r}: {bad.size} of {ref.shape[0]} anchors differ, first at {bad[0]}"
        n_ties += int((ref[1:, 0] == ref[:-1, 0]).sum())
    return n_ties


def test_anchor_lists_of_the_reference_map_o():
    """every read of tests/golden/ref_seed_hits.npz (reference test FASTA pairs + synthetic genome with repeats; four reads with equal x,
    where the order is radix_sort_128x's) through mm2c_seed_hits_batch_host, against what the reference's map.o gave mm_chain_dp"""
    import mm2chain
    d = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz"))
    reads = [(int(d[f"r{k}_qlen"]), d[f"r{k}_matches"], d[f"r{k}_hits"]) for k in range(int(d["n_reads"]))]
    mo, m, h, ql = _batch(reads)
    ao, a = mm2chain.seed_hits_batch(mo, m, h, ql)
    n_ties = 0
    for k in range(len(reads)):
        ref = d[f"r{k}_anchors"]
        assert np.array_equal(a[ao[k]:ao[k + 1]], ref), f"read {k} ({d[f'r{k}_src']}): anchors differ from the reference's"
        n_ties += int((ref[1:, 0] == ref[:-1, 0]).sum())
    assert n_ties > 1000


@pytest.mark.parametrize("seed", range(4))
def test_random_matches_against_the_oracle(seed):
    rng = np.random.default_rng(500 + seed)
    reads = [_random_read(rng, 900, 6, 3, 1 << 26),                       # no ties to speak of
             _random_read(rng, 700, 8, 2, 3000),                          # many equal x, several thousand anchors
             _random_read(rng, 40, 3, 1, 50),                             # <= 64 anchors with ties (insertion sort only)
             _random_read(rng, 0, 0, 1, 10),                              # a read without matches
             _random_read(rng, 300, 0, 1, 10),                            # matches without hits
             _random_read(rng, 1500, 5, 24, 1 << 27, dup_frac=0.3),       # duplicated hit lists: pairs of equal x across the genome
             _random_read(rng, 100, 2, 1, 1 << 20),
             _random_read(rng, 600, 4, 1 << 30, (1 << 31) - 1, dup_frac=0.3),   # x differs in more bits than the squeezed sort keys hold: anchors sorted as they are
             _random_read(rng, 2000, 3, 2, 200)]                          # nearly everything ties
    assert _check(reads, f"seed {seed}") > 1000


def test_reads_too_long_for_the_lds_replay():
    """the size classes of seed_ties beyond one wave per read: 12 289 .. 65 536 anchors (four waves, digits in LDS), 65 537 .. 131 072 (the same with
    128 KB of LDS) and beyond it (one wave, digits in global memory), with equal x"""
    rng = np.random.default_rng(77)
    reads = [_random_read(rng, 4500, 8, 2, 20000, qlen=60000), _random_read(rng, 3600, 10, 1, 1 << 24, qlen=60000, dup_frac=0.2),
             _random_read(rng, 7000, 8, 2, 50000, qlen=90000), _random_read(rng, 9000, 6, 3, 1 << 25, qlen=90000, dup_frac=0.25),
             _random_read(rng, 22000, 8, 2, 90000, qlen=150000, dup_frac=0.1), _random_read(rng, 38000, 8, 3, 1 << 23, qlen=200000, dup_frac=0.2)]
    sizes = [int(r[1]["n"].sum()) for r in reads]
    assert all(12288 < n <= 65536 for n in sizes[:4]), sizes
    assert 65536 < sizes[4] <= 131072 and sizes[5] > 131072, sizes
    assert _check(reads, "long") > 1000


def test_replay_with_the_digits_in_memory_on_one_two_four_and_eight_waves(monkeypatch):
    """reads beyond 131 072 anchors replay with their digits in memory, on one to eight waves per read by the number of such reads in the batch (the buckets of a level side by
    side; MM2C_TIE_GLOBAL_WAVES pins the number); MM2C_TIE_GLOBAL_ABOVE sends shorter reads the same way.  Every form against the oracle"""
    rng = np.random.default_rng(78)
    reads = [_random_read(rng, 38000, 8, 3, 1 << 23, qlen=200000, dup_frac=0.2), _random_read(rng, 2500, 8, 2, 30000, qlen=40000),
             _random_read(rng, 36000, 8, 1, 1 << 25, qlen=200000, dup_frac=0.1), _random_read(rng, 9000, 6, 3, 1 << 25, qlen=90000, dup_frac=0.25)]
    sizes = [int(r[1]["n"].sum()) for r in reads]
    assert sizes[0] > 131072 and sizes[2] > 131072, sizes
    ties = {}
    for waves in ("0", "1", "2", "4", "8"):
        monkeypatch.setenv("MM2C_TIE_GLOBAL_WAVES", waves)                        # read when a seed plan is made
        ties[waves] = _check(reads, f"MM2C_TIE_GLOBAL_WAVES={waves}")
    monkeypatch.setenv("MM2C_TIE_GLOBAL_WAVES", "2")
    monkeypatch.setenv("MM2C_TIE_GLOBAL_ABOVE", "5000")                           # the 9 000-match read (about 30 000 anchors) takes the same kernel
    ties["above"] = _check(reads, "MM2C_TIE_GLOBAL_ABOVE=5000")
    assert len(set(ties.values())) == 1 and ties["0"] > 1000, ties


@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("lo,hi,n_matches", [(12288, 16384, 3400), (6144, 12288, 2300)])
def test_reads_of_the_multi_wave_replay_classes(seed, lo, hi, n_matches):
    """reads with equal x whose replay runs level by level on several waves of a workgroup (replay_levels): 12 289 .. 16 384 anchors (the short end of the
    eight-wave class) and 6 145 .. 12 288 (two waves)"""
    rng = np.random.default_rng(900 + seed + lo)
    reads = []
    for pos_range, rids, dup in ((40000, 2, 0.0), (1 << 22, 3, 0.3), (3000, 1, 0.1)):
        while True:
            r = _random_read(rng, n_matches, 8, rids, pos_range, qlen=40000, dup_frac=dup)
            if lo < int(r[1]["n"].sum()) <= hi:
                break
        reads.append(r)
    assert _check(reads, f"mw {seed} {lo}") > 100


def test_lds_sort_and_global_sort_leave_the_same_anchor_lists(monkeypatch):
    """reads of up to 16 384 anchors whose differing x bits fit 32 are sorted in LDS (seed_sort_lds), the others through global memory (seed_sort); MM2C_LDS_SORT=0 sends every
    read through seed_sort.  Both routes against the oracle on the same reads: short and middle-sized ones, with and without equal x, one target and many (more than 32 differing bits)."""
    import mm2chain
    rng = np.random.default_rng(4242)
    reads = [_random_read(rng, 900, 6, 1, 1 << 22), _random_read(rng, 1200, 8, 2, 3000, dup_frac=0.2), _random_read(rng, 2600, 8, 3, 1 << 20, qlen=40000, dup_frac=0.1),
             _random_read(rng, 700, 5, 300, 1 << 30), _random_read(rng, 40, 3, 1, 500), _random_read(rng, 1500, 7, 1, 40000, dup_frac=0.3)]
    sizes = [int(r[1]["n"].sum()) for r in reads]
    assert max(sizes) > 5120 and min(sizes) < 200, sizes
    ties = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MM2C_LDS_SORT", flag)                                 # read when a seed plan is made
        ties[flag] = _check(reads, f"MM2C_LDS_SORT={flag}")
    assert ties["1"] == ties["0"] > 100


def test_anchor_offsets_that_do_not_match_the_hit_counts_are_reported():
    import mm2chain
    rng = np.random.default_rng(3)
    qlen, m, h = _random_read(rng, 50, 4, 1, 1 << 20)
    mo = np.array([0, m.size], np.int64)
    ao = np.array([0, int(m["n"].sum()) + 3], np.int64)
    plan = mm2chain.SeedPlan(mo, ao)
    d_m = torch.from_numpy(m.view(np.uint8)).cuda(); d_h = torch.from_numpy(h.view(np.int64)).cuda()
    d_q = torch.tensor([qlen], dtype=torch.int32, device="cuda")
    plan.run(d_m, d_h, d_q)
    with pytest.raises(mm2chain.Mm2cError):
        plan.check()
    plan.close()


def test_long_reads_expand_on_sixteen_waves_and_on_one_alike(monkeypatch):
    """reads beyond 16 384 anchors with every hit kept are expanded by the sixteen waves of a workgroup (seed_expand_mw: chunk totals, one scan, chunks at their places);
    MM2C_MW_SORT=0 leaves them to the one-wave kernels.  Both against the oracle: a match count that is no multiple of 64, long runs of matches without hits, one match with
    thousands of hits (a chunk far bigger than the others), next to short reads in the same batch"""
    rng = np.random.default_rng(6161)
    q1 = _random_read(rng, 9001, 6, 2, 1 << 24, qlen=120000, dup_frac=0.1)
    qlen, m, h = _random_read(rng, 6000, 9, 1, 1 << 25, qlen=100000)
    m = m.copy(); lists = [h[int(c):int(c) + int(n)] for c, n in zip(m["cr_off"], m["n"])]
    for k in range(1000, 2500):
        lists[k] = lists[k][:0]                                                     # 1 500 matches in a row without a hit
    big = np.sort(rng.integers(0, 1 << 25, 7000)).astype(np.uint64) << np.uint64(1)
    lists[4000] = big                                                               # one minimizer with 7 000 hits
    m["n"] = [x.size for x in lists]; m["cr_off"] = np.concatenate([[0], np.cumsum(m["n"].astype(np.int64))[:-1]])
    q2 = (qlen, m, np.concatenate(lists))
    reads = [_random_read(rng, 300, 5, 1, 1 << 20), q1, _random_read(rng, 0, 0, 1, 10), q2, _random_read(rng, 2500, 8, 2, 1 << 22, qlen=40000)]
    sizes = [int(r[1]["n"].sum()) for r in reads]
    assert sizes[1] > 16384 and sizes[3] > 16384 and max(sizes[0], sizes[4]) <= 16384, sizes
    ties = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MM2C_MW_SORT", flag)                                  # read when a seed plan is made
        ties[flag] = _check(reads, f"MM2C_MW_SORT={flag}")
    assert ties["1"] == ties["0"]


def test_anchor_offsets_of_a_long_read_that_do_not_match_are_reported():
    """the same check as above in the sixteen-wave expansion: the grand total of the chunk sums against the read's anchor range, before anything is written"""
    import mm2chain
    rng = np.random.default_rng(31)
    qlen, m, h = _random_read(rng, 6000, 8, 1, 1 << 24, qlen=90000)
    n = int(m["n"].sum())
    assert n > 16384
    for wrong in (n + 5, n - 5):
        plan = mm2chain.SeedPlan(np.array([0, m.size], np.int64), np.array([0, wrong], np.int64))
        d_m = torch.from_numpy(m.view(np.uint8)).cuda(); d_h = torch.from_numpy(h.view(np.int64)).cuda()
        d_q = torch.tensor([qlen], dtype=torch.int32, device="cuda")
        plan.run(d_m, d_h, d_q)
        with pytest.raises(mm2chain.Mm2cError):
            plan.check()
        plan.close()


def test_seeds_to_chains_without_leaving_the_device():
    """matches -> anchors -> f/p -> chains on the GPU (SeedPlan, ChainPlan.run, ChainPlan.chains on one stream) against the oracle's
    collect_seed_hits + mm_chain_dp per read"""
    import mm2chain
    from mm2chain import params
    d = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz"))
    reads = [(int(d[f"r{k}_qlen"]), d[f"r{k}_matches"], d[f"r{k}_hits"]) for k in range(int(d["n_reads"]))]
    mo, m, h, ql = _batch(reads)
    ao = np.concatenate([[0], np.cumsum([int(r[1]["n"].sum()) for r in reads])]).astype(np.int64)
    P = params.map_ont()
    sp = mm2chain.SeedPlan(mo, ao)
    cp = mm2chain.ChainPlan(P, ao)
    d_a = sp.run(torch.from_numpy(m.view(np.uint8)).cuda(), torch.from_numpy(h.view(np.int64)).cuda(), torch.from_numpy(ql).cuda())
    d_f = torch.empty(int(ao[-1]), dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
    cp.run(d_a, d_f, d_p)
    u_off, u, b_off, b = cp.chains(d_a, d_f, d_p, 3, 40)
    torch.cuda.synchronize()
    assert sp.check() >= 4 and sp.last_ms() > 0
    uo, bo = u_off.cpu().numpy(), b_off.cpu().numpy()
    u, b = u.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64)
    for k in range(len(reads)):
        u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, d[f"r{k}_anchors"])
        assert np.array_equal(u[uo[k]:uo[k + 1]], u_ref) and np.array_equal(b[bo[k]:bo[k + 1]], b_ref), f"read {k}: chains differ"
    sp.close(); cp.close()


def test_matches_in_chains_out_host_entry():
    """mm2c_seed_chain_batch_host on the fixture: chains of every read against the oracle's mm_chain_dp on the reference's anchors"""
    import mm2chain
    from mm2chain import params
    d = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz"))
    reads = [(int(d[f"r{k}_qlen"]), d[f"r{k}_matches"], d[f"r{k}_hits"]) for k in range(int(d["n_reads"]))]
    mo, m, h, ql = _batch(reads)
    P = params.map_ont()
    res = mm2chain.seed_chain_batch(P, 3, 40, mo, m, h, ql)
    n_chains = 0
    for k in range(len(reads)):
        u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, d[f"r{k}_anchors"])
        assert np.array_equal(res[k][0], u_ref) and np.array_equal(res[k][1], b_ref), f"read {k}: chains differ"
        n_chains += u_ref.size
    assert n_chains > 27


@pytest.mark.parametrize("mode", ["host-pool-ranges", "host-pool-shared", "resident-pool"])
def test_matches_in_chains_out_pipelined_in_chunks(mode):
    """the same entry with a batch big enough (here: a chunk size small enough) to run as a two-stream pipeline of chunks of whole reads:
    chunk plans from the device cache, hits uploaded per chunk as the range the chunk's matches point into ("host-pool-ranges"), once as a whole
    when every chunk points all over the pool ("host-pool-shared": the reads are shuffled against the pool), or not at all
    ("resident-pool": mm2c_hitpool_create + mm2c_seed_chain_batch_pool).  Chains of every read against the oracle; stage statistics populated."""
    import mm2chain
    from mm2chain import params
    d = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz"))
    base = [(int(d[f"r{k}_qlen"]), d[f"r{k}_matches"], d[f"r{k}_hits"], d[f"r{k}_anchors"]) for k in range(int(d["n_reads"]))]
    reads = base * 6                                                # 162 reads
    mo, m, h, ql = _batch([r[:3] for r in reads])
    if mode == "host-pool-shared":
        # the same reads in another order than their hit lists lie in the pool: every chunk's matches then span the whole pool
        order = np.random.default_rng(5).permutation(len(reads))
        cnt = np.diff(mo)
        m = np.concatenate([m[mo[k]:mo[k + 1]] for k in order])
        mo = np.concatenate([[0], np.cumsum(cnt[order])]).astype(np.int64)
        ql = ql[order]
        reads = [reads[k] for k in order]
    P = params.map_ont()
    total = sum(int(r[1]["n"].sum()) for r in reads)
    mm2chain.tune("pipeline_chunk_anchors", max(1024, total // 7))
    mm2chain.stage_stats(reset=True)
    try:
        if mode == "resident-pool":
            pool = mm2chain.HitPool(h)
            res = mm2chain.seed_chain_batch_pool(P, 3, 40, mo, m, pool, ql)
            res2 = mm2chain.seed_chain_batch_pool(P, 3, 40, mo, m, pool, ql)        # arenas and cached plan workspace reused
            pool.close()
        else:
            res = mm2chain.seed_chain_batch(P, 3, 40, mo, m, h, ql)
            res2 = mm2chain.seed_chain_batch(P, 3, 40, mo, m, h, ql)
    finally:
        mm2chain.tune("pipeline_chunk_anchors", 20 << 20)
    st = mm2chain.stage_stats()
    assert st["calls"] == 2 and st["chunks"] >= 10 and st["seed_ns"] > 0 and st["dp_ns"] > 0 and st["epi_ns"] > 0 and st["total_ns"] > 0, st
    for k, r in enumerate(reads):
        u_ref, b_ref = ob.mm_chain_dp(P, 3, 40, r[3])
        assert np.array_equal(res[k][0], u_ref) and np.array_equal(res[k][1], b_ref), f"{mode}: read {k}: chains differ"
        assert np.array_equal(res2[k][0], u_ref) and np.array_equal(res2[k][1], b_ref), f"{mode}: second call, read {k}: chains differ"


def test_all_vs_all_seed_hits_with_skip_seed_equal_the_reference_and_chain_on_the_device():
    """`-x ava-ont` (BASELINE config 5; options.c:82-86: NO_DIAG | NO_DUAL): the matches of 37 reads mapped against themselves go through
    mm2c_seedplan_run_device_skip -- skip_seed (map.c:122-147) with the name comparison as ranks, MM_SEED_SELF (map.c:241), reads that keep
    fewer anchors than they have hits, packed by offsets the device computes -- and must equal, read by read, the anchor lists the
    reference's own map.o handed to mm_chain_dp (tests/golden/ref_seed_hits_ava.npz).  Then the anchors are chained where they are:
    plan over the capacity offsets + mm2c_plan_set_device_offsets; f / p against the oracle on the reference's anchors."""
    import mm2chain
    from mm2chain import params
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_seed_hits_ava.npz"))
    n = int(d["n_reads"])
    ms, hs, mo, cap = [], [], [0], [0]
    for k in range(n):
        m = d[f"r{k}_matches"].copy(); m["cr_off"] += cap[-1]
        ms.append(m); hs.append(d[f"r{k}_hits"]); mo.append(mo[-1] + m.size); cap.append(cap[-1] + int(m["n"].sum()))
    m_all = np.concatenate(ms); h_all = np.concatenate(hs)
    if True:
        sp = mm2chain.SeedPlan(np.array(mo, np.int64), np.array(cap, np.int64))
        dev = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).astype(dt)).cuda()
        d_m = torch.from_numpy(m_all.view(np.uint8).copy()).cuda(); d_h = torch.from_numpy(h_all.view(np.int64)).cuda()
        d_q = dev([int(d[f"r{k}_qlen"]) for k in range(n)], np.int32)
        d_lo = dev([int(d[f"r{k}_qlo"]) for k in range(n)], np.int32); d_eq = dev([int(d[f"r{k}_qeq"]) for k in range(n)], np.int32)
        d_rr = dev(d["ref_rank"], np.int32); d_rl = dev(d["ref_len"], np.int32)
        anchors, off = sp.run_skip(d_m, d_h, d_q, int(d["flag"]), d_rr, d_rl, d_lo, d_eq)
        sp.check()
        off_h = off.cpu().numpy()
        a_h = anchors.cpu().numpy().view(np.uint64)
        assert off_h[0] == 0
        n_self = 0
        for k in range(n):
            ref = d[f"r{k}_anchors"]
            got = a_h[off_h[k]:off_h[k + 1]]
            assert np.array_equal(got, ref), f"read {k}: {got.shape[0]} anchors, the reference kept {ref.shape[0]}"
            n_self += int(((ref[:, 1] >> np.uint64(43)) & np.uint64(1)).sum())
        assert n_self > 20 and int(off_h[-1]) < cap[-1] // 4          # most hits are dropped (the diagonal and the second copy of every pair)
        # the flag-free entry on the same plan still keeps everything
        a2 = sp.run(d_m, d_h, d_q); sp.check()
        assert np.array_equal(a2.cpu().numpy().view(np.uint64)[:cap[1]], ob.collect_seed_hits(d["r0_matches"], d["r0_hits"], int(d["r0_qlen"])))
        # ---- chain them where they are, with the sizes only the device knows
        h = [int(v) for v in d["chain_scalars"]]
        P = params.make_params(h[0], h[1], h[2], h[3], h[4], 1.0, h[7], h[8])
        plan = mm2chain.ChainPlan(P, np.array(cap, np.int64))
        plan.set_device_offsets(off)
        d_f = torch.full((cap[-1],), -9, dtype=torch.int32, device="cuda"); d_p = torch.full_like(d_f, -9)
        plan.run(anchors, d_f, d_p)
        u_off, u, b_off, b = plan.chains(anchors, d_f, d_p, h[5], h[6])
        f_h, p_h = d_f.cpu().numpy(), d_p.cpu().numpy()
        uo, bo = u_off.cpu().numpy(), b_off.cpu().numpy()
        u_h, b_h = u.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64)
        n_chains = 0
        for k in range(n):
            ref = d[f"r{k}_anchors"]
            if ref.shape[0] == 0:
                assert uo[k + 1] == uo[k]
                continue
            f_ref, p_ref, _ = ob.chain_fpv(P, ref)
            assert np.array_equal(f_h[off_h[k]:off_h[k + 1]], f_ref) and np.array_equal(p_h[off_h[k]:off_h[k + 1]], p_ref), k
            u_ref, b_ref = ob.mm_chain_dp(P, h[5], h[6], ref)
            assert np.array_equal(u_h[uo[k]:uo[k + 1]], u_ref) and np.array_equal(b_h[bo[k]:bo[k + 1]], b_ref), k
            n_chains += u_ref.size
        assert n_chains > 20
        plan.close(); sp.close()


def _run_heap_plan(reads, skip=None):
    """reads through a seed plan with mm2c_seedplan_set_heap_sort(1); skip = (flag, ref_rank, ref_len, [q_lo], [q_eq]) for the skip_seed entry.
    Returns per-read anchor arrays."""
    import mm2chain
    mo, m, h, ql = _batch(reads)
    cap = np.concatenate([[0], np.cumsum([int(r[1]["n"].sum()) for r in reads])]).astype(np.int64)
    sp = mm2chain.SeedPlan(mo, cap)
    sp.set_heap_sort(True)
    d_m = torch.from_numpy(m.view(np.uint8).copy()).cuda(); d_h = torch.from_numpy(h.view(np.int64).copy()).cuda(); d_q = torch.from_numpy(ql).cuda()
    if skip is None:
        a = sp.run(d_m, d_h, d_q); sp.check()
        off = cap
    else:
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).astype(np.int32)).cuda()
        a, off = sp.run_skip(d_m, d_h, d_q, skip[0], dev(skip[1]), dev(skip[2]), dev(skip[3]), dev(skip[4])); sp.check()
        off = off.cpu().numpy()
    a = a.cpu().numpy().view(np.uint64)
    out = [a[off[k]:off[k + 1]] for k in range(len(reads))]
    # the plan goes back to the radix sort's order when the switch is turned off
    sp.set_heap_sort(False)
    if skip is None:
        a2 = sp.run(d_m, d_h, d_q); sp.check()
        assert np.array_equal(a2.cpu().numpy().view(np.uint64)[:cap[1]], ob.collect_seed_hits(reads[0][1], reads[0][2], reads[0][0]))
    sp.close()
    return out


def test_heap_sort_anchor_lists_of_the_reference_map_o():
    """MM_F_HEAP_SORT (--heap-sort, main.c:245; -x sr, options.c:125): collect_seed_hits_heap (map.c:149-213) merges the matches' hit lists through a binary
    heap, which leaves anchors with equal x in another order than radix_sort_128x.  Every read of ref_seed_hits.npz through a plan with
    mm2c_seedplan_set_heap_sort against the anchor lists the reference's own map.o handed to mm_chain_dp under that flag
    (tests/golden/ref_seed_hits_heap.npz, make_ref_heap_fixtures.py); in four of the reads anchors sit at other places than in the radix-sorted list"""
    d, hp = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz")), np.load(os.path.join(GOLDEN, "ref_seed_hits_heap.npz"))
    reads = [(int(d[f"r{k}_qlen"]), d[f"r{k}_matches"], d[f"r{k}_hits"]) for k in range(int(d["n_reads"]))]
    got = _run_heap_plan(reads)
    n_moved = 0
    for k in range(len(reads)):
        ref = hp[f"r{k}_anchors_heap"]
        assert np.array_equal(got[k], ref), f"read {k}: anchors differ from the reference's heap-merged list"
        n_moved += int((ref != d[f"r{k}_anchors"]).any(axis=1).sum())
    assert n_moved > 2000


def _sorted_lists(read):
    """the hit lists of a random read in ascending order, as mm_idx_get hands them out (the heap merge presupposes it)"""
    qlen, m, h = read
    h = h.copy()
    for k in range(m.size):
        lo, n = int(m["cr_off"][k]), int(m["n"][k])
        h[lo:lo + n] = np.sort(h[lo:lo + n])
    return qlen, m, h


@pytest.mark.parametrize("seed", range(3))
def test_heap_sort_random_matches_against_the_oracle(seed):
    """random matches with many equal x (small position ranges, duplicated hit lists), more matches than the LDS heap holds, reads without hits:
    the replayed heap against the oracle's (mm2o_collect_seed_hits_heap), with and without skip_seed flags"""
    rng = np.random.default_rng(900 + seed)
    reads = [_sorted_lists(r) for r in (_random_read(rng, 700, 8, 2, 3000), _random_read(rng, 40, 3, 1, 50), _random_read(rng, 0, 0, 1, 10),
                                        _random_read(rng, 300, 0, 1, 10), _random_read(rng, 1500, 5, 24, 1 << 27, dup_frac=0.3),
                                        _random_read(rng, 2600, 3, 2, 400), _random_read(rng, 900, 6, 3, 1 << 26), _random_read(rng, 2000, 3, 2, 200))]
    got = _run_heap_plan(reads)
    n_moved = 0
    for k, (qlen, m, h) in enumerate(reads):
        ref = ob.collect_seed_hits(m, h, qlen, heap=True)
        assert np.array_equal(got[k], ref), f"seed {seed}, read {k}: heap order differs from the oracle's"
        n_moved += int((ref != ob.collect_seed_hits(m, h, qlen)).any(axis=1).sum())
    assert n_moved > 500
    # strand-restricted and all-vs-all flags: fewer anchors than hits (rid 0 .. 23 here; every read named like reference 1, of the read's length)
    ref_rank, ref_len = np.arange(24, dtype=np.int32), np.full(24, 12000, np.int32)
    for flag in (ob.F_FOR_ONLY, ob.F_NO_DIAG | ob.F_NO_DUAL):
        q_lo, q_eq = np.ones(len(reads), np.int32), np.ones(len(reads), np.int32)
        got = _run_heap_plan(reads, skip=(flag, ref_rank, ref_len, q_lo, q_eq))
        for k, (qlen, m, h) in enumerate(reads):
            ref = ob.collect_seed_hits(m, h, qlen, flag, ref_rank, ref_len, 1, 1, heap=True)
            assert np.array_equal(got[k], ref), f"seed {seed}, flag {flag:#x}, read {k}: {got[k].shape[0]} anchors vs {ref.shape[0]}"


def test_heap_sort_through_the_host_batch_entry():
    """mm2c_tune("heap_sort", 1): the seed plans the host-batch entries create themselves leave the heap order too (a host that maps with MM_F_HEAP_SORT)"""
    import mm2chain
    d, hp = np.load(os.path.join(GOLDEN, "ref_seed_hits.npz")), np.load(os.path.join(GOLDEN, "ref_seed_hits_heap.npz"))
    reads = [(int(d[f"r{k}_qlen"]), d[f"r{k}_matches"], d[f"r{k}_hits"]) for k in range(int(d["n_reads"]))]
    mo, m, h, ql = _batch(reads)
    mm2chain.tune("heap_sort", 1)
    try:
        ao, a = mm2chain.seed_hits_batch(mo, m, h, ql)
    finally:
        mm2chain.tune("heap_sort", 0)
    for k in range(len(reads)):
        assert np.array_equal(a[ao[k]:ao[k + 1]], hp[f"r{k}_anchors_heap"]), k
    ao, a = mm2chain.seed_hits_batch(mo, m, h, ql)
    assert all(np.array_equal(a[ao[k]:ao[k + 1]], d[f"r{k}_anchors"]) for k in range(len(reads)))
