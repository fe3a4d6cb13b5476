#!/usr/bin/env python3
"""Parity soak (GPU box): many random parameter sets x adversarial / synthetic anchor lists, HIP path vs CPU oracle, bit for bit.
usage: tools/soak.py [n_rounds] [seed0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, mm2chain
from mm2chain import params, synth
from helpers import mk_anchor, pack, oracle_batch, gpu_batch, respan_q
from test_gpu_parity import _random_task
INT32_MAX = 2**31 - 1
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mm2chain.init()
if os.environ.get("MM2C_SOAK_CUT"):          # exercise the device-side cut of plans on small tasks as well
    mm2chain.tune("plan_cut_min", 100); mm2chain.tune("seg_min", int(os.environ["MM2C_SOAK_CUT"]))
t0 = time.time(); n_anchor = 0; bad = 0
budget = float(os.environ.get("MM2C_SOAK_SECONDS", "0")); t_last = time.time()
for r in range(rounds):
    if budget and time.time() - t0 > budget:
        rounds = r
        break
    if time.time() - t_last > 60:
        t_last = time.time(); print(f"... round {r}, {time.time() - t0:.0f} s", flush=True)
    rng = np.random.default_rng(seed0 + r)
    n_segs = int(rng.choice([1, 1, 1, 2, 3]))
    P = params.make_params(max_dist_x=int(rng.choice([0, 50, 700, 5000, 10000, 100000])), max_dist_y=int(rng.choice([-5, 60, 700, 5000, 10000])),
                           bw=int(rng.choice([-1, 0, 10, 500, 2000, 5000])), max_skip=int(rng.choice([-1, 0, 1, 5, 25, 25, 300, INT32_MAX])),
                           max_iter=int(rng.choice([-3, 0, 1, 63, 64, 65, 200, 1024, 5000, 5000, INT32_MAX])),
                           gap_scale=float(rng.choice([1.0, 1.0, 0.5, 0.8, 2.25, 0.0])), is_cdna=int(rng.integers(0, 2)) if n_segs > 1 or rng.random() < .2 else 0,
                           n_segs=n_segs)
    mm2chain.tune("far_ring", int(rng.choice([1, 1, 2, 0])))   # ring-size classes of the tile kernel: chosen per task, all long, all short
    mm2chain.tune("q24_ring", int(rng.choice([1, 1, 1, 0])))       # the long ring in its q24 form (round 5) or with 32-bit slots
    mm2chain.tune("compact_ring", int(rng.choice([1, 1, 1, 0])))   # the compact x / q ring for the tasks whose q values allow it, or never
    mm2chain.tune("wide_share_threshold", int(rng.choice([100, 100, 40, 0])))   # ... and the share of anchors in 32-bit-ring tasks from which every task takes that ring
    mm2chain.tune("split_streams", int(rng.choice([1, 2, 2, 0])))
    mm2chain.tune("noskip_loop", int(rng.choice([1, 1, 0])))     # max_skip >= max_iter: through the hand-written loop with max_skip = max_iter - 1, or the C++ loop without the machinery
    mm2chain.tune("ring_class", int(rng.choice([3, 3, 3, 3, 4, 4, 0, 1, 2])))   # mostly the tile kernel (the default), sometimes the first-generation one
    # several waves per task (chain_dp_coop.h): always with MM2C_SOAK_COOP set, else in a third of the rounds
    # round 6: gpu_batch runs every input on BOTH routes (one wave per piece, and the library's default, which sends few long pieces -- most soak batches -- to the
    # cooperative kernel, through the device-side route when tasks are cut first) and compares them; in a third of the rounds the cooperative kernel is forced instead
    import helpers
    if os.environ.get("MM2C_SOAK_COOP") or rng.random() < 0.33:
        helpers.PINNED_ROUTE = 1; mm2chain.tune("coop_plans", 1)
    else:
        helpers.PINNED_ROUTE = None; mm2chain.tune("coop_plans", 2)
    mm2chain.tune("coop_w8_above", int(rng.choice([256, 256, 0, 3])))   # the cooperative kernel's width: sixteen waves per piece, eight beyond this many pieces
    mm2chain.tune("single_launch", int(rng.choice([1, 1, 0])))    # per-read passes: the cooperative kernel reads the pinned arena itself, or stage_in uploads it first
    mm2chain.tune("fuse_st", int(rng.choice([1, 1, 0])))          # short tasks in the sixteen-wave kernel: window starts made by the kernel itself, or by a prepass launch
    mm2chain.tune("fused_out", int(rng.choice([1, 1, 0])))       # per-read passes: the cooperative kernel writes the caller's buffer and raises the flag, or stage_out does
    mm2chain.tune("host_st", int(rng.choice([0, 0, 1])))         # ... and their window starts from the prepass kernel or from the host
    tasks = []
    for _ in range(int(rng.integers(1, 12))):
        kind = rng.random()
        if kind < 0.08:      # long dense tasks: windows of several hundred anchors (look-back beyond the LDS rings), pieces cut on the device
            _, a = synth.make_stream("dense", 1, int(rng.integers(3000, 14000)), seed=int(rng.integers(0, 1 << 30)), locus=int(rng.choice([1500, 3000, 9000])))
            tasks.append(a.numpy().view(np.uint64))
        elif kind < 0.5:
            tasks.append(_random_task(rng, int(rng.integers(1, 2500)), int(rng.integers(1, 4)), n_segs, bool(rng.integers(0, 2))))
        else:
            prof = str(rng.choice(["mixed", "dense", "colinear", "sparse"]))
            _, a = synth.make_stream(prof, 1, int(rng.integers(1, 3000)), seed=int(rng.integers(0, 1 << 30)), q_span=int(rng.choice([15, 19])),
                                     locus=int(rng.choice([3000, 20000, 100000])) if prof != "sparse" else None)
            tasks.append(a.numpy().view(np.uint64))
    tasks = [respan_q(rng, t, min(P.max_dist_x, P.max_dist_y), int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7]))) for t in tasks]   # corners of the compact ring's bound on q
    # the q24 ring's bound: now and then a task's q values are moved so that the largest sits at 2^24 - 1 or a little beyond (that task must stay out of the long ring)
    for k in range(len(tasks)):
        if tasks[k].shape[0] and rng.random() < 0.15:
            t = tasks[k].copy()
            q = (t[:, 1] & np.uint64(0xffffffff)).astype(np.int64)
            q = q + ((1 << 24) - 1 + int(rng.integers(-2, 3)) - int(q.max()))
            if q.min() >= 0:
                t[:, 1] = (t[:, 1] & np.uint64(0xffffffff00000000)) | q.astype(np.uint64)
                tasks[k] = t
    a = np.concatenate(tasks); off = np.concatenate(([0], np.cumsum([t.shape[0] for t in tasks]))).astype(np.int64)
    f_ref, p_ref = oracle_batch(P, off, a)
    f, p = gpu_batch(P, off, a)
    # the host-buffer entries too (round 5: their small passes stage through kernels and a polled flag, csrc/host_stage.hip): the batch entry and a per-read call
    if P.max_dist_x >= 0 and rng.random() < 0.5:
        fh, ph = mm2chain.chain_batch_host(P, off, a)
        if not (np.array_equal(fh, f_ref) and np.array_equal(ph, p_ref)):
            bad += 1; print(f"MISMATCH (host batch entry) round {r} seed {seed0 + r}")
        k = int(rng.integers(0, len(tasks)))
        if tasks[k].shape[0]:
            import oracle_binding as ob
            ft, pt = mm2chain.chain_task(P, tasks[k], ob.avg_qspan(tasks[k]), tid=int(rng.integers(0, 16)))
            if not (np.array_equal(ft, f_ref[off[k]:off[k + 1]]) and np.array_equal(pt, p_ref[off[k]:off[k + 1]])):
                bad += 1; print(f"MISMATCH (per-read call) round {r} seed {seed0 + r} task {k}")
    n_anchor += a.shape[0]
    if not (np.array_equal(f, f_ref) and np.array_equal(p, p_ref)):
        bad += 1
        i = int(np.nonzero((f != f_ref) | (p != p_ref))[0][0])
        print(f"MISMATCH round {r} seed {seed0 + r}: first at {i}: f {f[i]} vs {f_ref[i]}, p {p[i]} vs {p_ref[i]}; params {params.as_dict(P)}")
mm2chain.tune("coop_plans", 2); mm2chain.tune("coop_w8_above", 256); mm2chain.tune("fuse_st", 1); mm2chain.tune("single_launch", 1); mm2chain.tune("fused_out", 1); mm2chain.tune("host_st", 0)
mm2chain.tune("ring_class", 3); mm2chain.tune("far_ring", 1); mm2chain.tune("compact_ring", 1); mm2chain.tune("q24_ring", 1); mm2chain.tune("wide_share_threshold", 40); mm2chain.tune("split_streams", 1); mm2chain.tune("noskip_loop", 1)
print(f"soak: {rounds} rounds, {n_anchor} anchors, {bad} mismatching rounds, {time.time() - t0:.1f} s")
