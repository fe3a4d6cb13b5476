#!/usr/bin/env python3
"""randomised parity soak of the kernels around the DP: device epilogue (f/p -> chains) against the host epilogue on arbitrary forests and
on real DP outputs with random thresholds, and seed hits -> anchors against the oracle on random match lists.
usage: python tools/soak2.py [rounds] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import mm2chain
from mm2chain import params, synth
import oracle_binding as ob

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mm2chain.init()
P = params.map_ont()
t0 = time.time()
n_forest = n_dp = n_seed = 0
anchors_done = 0


def forest(rng, n, k):
    style = int(rng.integers(0, 5))
    idx = np.arange(n)
    if style == 0:
        par = np.where(rng.random(n) < rng.uniform(0.8, 0.999), idx - 1, -1)
    elif style == 1:
        par = np.where(rng.random(n) < 0.8, (rng.random(n) * idx).astype(np.int64) - (idx == 0), -1)
    elif style == 2:
        par = idx - 1 - rng.integers(0, int(rng.integers(1, 9)), n)
    elif style == 3:
        par = np.where(rng.random(n) < 0.5, idx - 1, idx - 1 - rng.integers(0, 300, n))
    else:
        par = np.full(n, -1)
    par = np.where(par < 0, -1, par).astype(np.int32)
    mode = int(rng.integers(0, 3))
    if mode == 0:
        f = rng.integers(0, 80, n)
    else:
        gain = rng.integers(-30, 26 if mode == 1 else 16, n)
        f = np.zeros(n, np.int64)
        for i in range(n):
            f[i] = max(15, (f[par[i]] if par[i] >= 0 else 0) + 15 + gain[i])
    x = np.sort(rng.integers(0, int(rng.choice([40, 3000, 1 << 30])), n).astype(np.uint64)) + (np.uint64(k) << np.uint64(32))
    a = np.zeros((n, 2), np.uint64); a[:, 0] = x
    a[:, 1] = (np.uint64(15) << np.uint64(32)) | rng.integers(0, 1 << 20, n).astype(np.uint64)
    return a, f.astype(np.int32), par


def check_chains(tag, off, got, ref):
    for k in range(off.size - 1):
        if not (np.array_equal(got[k][0], ref[k][0]) and np.array_equal(got[k][1], ref[k][1])):
            print(f"MISMATCH {tag}: task {k} n={off[k+1]-off[k]}: {got[k][0].size} vs {ref[k][0].size} chains"); sys.exit(1)


def device_chains(off, a, f, p, min_cnt, min_sc):
    plan = mm2chain.ChainPlan(P, off)
    u_off, u, b_off, b = plan.chains(torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(f).cuda(), torch.from_numpy(p).cuda(), min_cnt, min_sc)
    torch.cuda.synchronize()
    uo, bo = u_off.cpu().numpy(), b_off.cpu().numpy()
    u, b = u.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64)
    plan.close()
    return [(u[uo[k]:uo[k + 1]], b[bo[k]:bo[k + 1]]) for k in range(off.size - 1)]


budget = float(os.environ.get("MM2C_SOAK_SECONDS", "0")); t_last = time.time()
for r in range(rounds):
    if budget and time.time() - t0 > budget:
        rounds = r
        break
    if time.time() - t_last > 60:
        t_last = time.time(); print(f"... round {r}, {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(seed0 * 100003 + r)
    # 1. arbitrary forests
    sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, int(rng.integers(1, 9000))])) for _ in range(int(rng.integers(1, 12)))]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    parts = [forest(rng, n, k) for k, n in enumerate(sizes) if n > 0]
    if parts:
        a = np.concatenate([x[0] for x in parts]); f = np.concatenate([x[1] for x in parts]); p = np.concatenate([x[2] for x in parts])
        min_cnt, min_sc = int(rng.integers(0, 6)), int(rng.choice([-3, 0, 15, 30, 40, 100, 400]))
        check_chains(f"forest round {r}", off, device_chains(off, a, f, p, min_cnt, min_sc), mm2chain.chain_epilogue_host(min_cnt, min_sc, off, a, f, p, n_threads=4))
        n_forest += 1; anchors_done += a.shape[0]
    # 2. real DP output, whole-function entry, random thresholds and profile
    prof = str(rng.choice(["mixed", "dense", "sparse", "colinear"]))
    off_t, a_t = synth.make_stream(prof, int(rng.integers(1, 10)), (1, int(rng.integers(2, 4000))), seed=int(rng.integers(1, 1 << 30)))
    off2, a2 = off_t.numpy(), a_t.numpy().view(np.uint64)
    min_cnt, min_sc = int(rng.integers(1, 5)), int(rng.choice([0, 20, 40, 100]))
    got = mm2chain.mm_chain_dp_batch(P, min_cnt, min_sc, off2, a2, epilogue_threads=0)
    for k in range(off2.size - 1):
        u_ref, b_ref = ob.mm_chain_dp(P, min_cnt, min_sc, a2[off2[k]:off2[k + 1]])
        if not (np.array_equal(got[k][0], u_ref) and np.array_equal(got[k][1], b_ref)):
            print(f"MISMATCH dp round {r} task {k} profile {prof}"); sys.exit(1)
    n_dp += 1; anchors_done += a2.shape[0]
    # 3. seed hits
    reads = []
    for _ in range(int(rng.integers(1, 6))):
        nm = int(rng.integers(0, 1500)) if rng.random() > 0.12 else int(rng.integers(3000, 24000))   # now and then a read of the longer size classes of seed_ties (up to ~10^5 anchors)
        max_n = int(rng.integers(0, 9))
        m = np.zeros(nm, ob.MATCH_DTYPE)
        qlen = int(rng.integers(100, 50000))
        m["q_pos"] = (np.sort(rng.integers(15, qlen, nm)).astype(np.uint32) << 1) | rng.integers(0, 2, nm).astype(np.uint32)
        m["q_span"] = rng.integers(1, 32, nm); m["seg_tandem"] = rng.integers(0, 8, nm)
        pos_range = int(rng.choice([30, 500, 70000, 1 << 22, 1 << 30])); rids = int(rng.choice([1, 2, 300]))
        lists = []
        for k in range(nm):
            n = int(rng.integers(0, max_n + 1))
            if lists and rng.random() < 0.2 and lists[-1].size == n:
                lists.append(lists[-1].copy()); continue
            lists.append((rng.integers(0, rids, n).astype(np.uint64) << np.uint64(32)) | (np.sort(rng.integers(0, pos_range, n)).astype(np.uint64) << np.uint64(1)) | rng.integers(0, 2, n).astype(np.uint64))
        m["n"] = [x.size for x in lists]
        m["cr_off"] = np.concatenate([[0], np.cumsum(m["n"].astype(np.int64))[:-1]]) if nm else np.zeros(0, np.int64)
        reads.append((qlen, m, np.concatenate(lists) if lists else np.zeros(0, np.uint64)))
    mo, ms, hs, ql, base = [0], [], [], [], 0
    for qlen, m, h in reads:
        mm = m.copy(); mm["cr_off"] += base; base += h.size
        ms.append(mm); hs.append(h); ql.append(qlen); mo.append(mo[-1] + m.size)
    ao, ag = mm2chain.seed_hits_batch(np.array(mo, np.int64), np.concatenate(ms), np.concatenate(hs), np.array(ql, np.int32))
    for k, (qlen, m, h) in enumerate(reads):
        if not np.array_equal(ag[ao[k]:ao[k + 1]], ob.collect_seed_hits(m, h, qlen)):
            print(f"MISMATCH seeds round {r} read {k}"); sys.exit(1)
    n_seed += 1; anchors_done += int(ao[-1])
    if r % 20 == 19:
        print(f"round {r + 1}: ok ({anchors_done} anchors, {time.time() - t0:.0f} s)", flush=True)
print(f"soak2: {rounds} rounds ({n_forest} forest batches, {n_dp} DP batches, {n_seed} seed batches, {anchors_done} anchors), 0 mismatches, {time.time() - t0:.0f} s")
mm2chain.shutdown()
