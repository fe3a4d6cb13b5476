#!/usr/bin/env python3
"""Latency of ONE synchronous chaining call (the reference's call pattern, chain.c:103 -> run_chaining_on_hw) on a lone bench-stream task:
mm2c_chain_task_host (V1 scalars) and the run_chaining_on_hw symbol (V2), per stream profile, best and median of 50 calls."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, mm2chain
from mm2chain import params, synth
import oracle_binding as ob
mm2chain.init()
for name, P, prof, n, locus, qs in (("map-ont mixed", params.map_ont(), "mixed", 5000, None, 15), ("map-ont dense", params.map_ont(), "dense", 5000, None, 15),
                                    ("map-ont colinear", params.map_ont(), "colinear", 5000, None, 15), ("map-ont sparse", params.map_ont(), "sparse", 5000, None, 15),
                                    ("ava-ont mixed", params.ava_ont(), "mixed", 20000, 400000, 15), ("tiny", params.map_ont(), "mixed", 8, None, 15)):
    off, a = synth.make_stream(prof, 1, n, seed=1, q_span=qs, locus=locus)
    t = a.numpy().view(np.uint64)
    avg = ob.avg_qspan(t)
    f_ref, p_ref, _ = ob.chain_fpv(P, t, avg)
    P2 = params.make_params(P.max_dist_x, P.max_dist_y, P.bw, max_skip=2**31 - 1, max_iter=1024, q_span_override=qs, flags=mm2chain.MM2C_F_IGNORE_SEG)
    f2_ref, p2_ref, _ = ob.chain_fpv(P2, t, avg)
    for coop in (0, 16):
        mm2chain.tune("coop_waves", coop)
        ts = []
        for k in range(60):
            t0 = time.perf_counter(); f, p = mm2chain.chain_task(P, t, avg); ts.append(time.perf_counter() - t0)
        ok = np.array_equal(f, f_ref) and np.array_equal(p, p_ref)
        ts = np.array(ts[10:]) * 1e3
        print(f"{name:18s} n={n:6d} waves per piece {max(coop, 1):2d}: mm2c_chain_task_host (V1) best {ts.min():.3f} ms median {np.median(ts):.3f} ms  identical={ok}  [{mm2chain.last_host_variant()[:40]}]")
        ts = []
        for k in range(60):
            t0 = time.perf_counter(); r, f, p = mm2chain.run_chaining_on_hw(n, P.max_dist_x, P.max_dist_y, P.bw, qs, avg, t); ts.append(time.perf_counter() - t0)
        ok = np.array_equal(f, f2_ref) and np.array_equal(p, p2_ref)
        ts = np.array(ts[10:]) * 1e3
        print(f"{name:18s} n={n:6d} waves per piece {max(coop, 1):2d}: run_chaining_on_hw (V2)    best {ts.min():.3f} ms median {np.median(ts):.3f} ms  identical={ok}")
    t0 = time.perf_counter()
    for k in range(5): ob.chain_fpv(P, t, avg)
    print(f"{name:18s} CPU oracle 1 thread: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
