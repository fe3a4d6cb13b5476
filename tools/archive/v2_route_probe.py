#!/usr/bin/env python3
"""V2 scalars (what run_chaining_on_hw passes: max_skip = INT_MAX, max_iter = 1024) through the C++ loop of the tile kernel (SKIP = false) and, with max_skip = max_iter - 1
(same results: the counter cannot pass max_iter - 1), through the hand-written loop: DP kernel ms on the bench's streams."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import numpy as np, torch, mm2chain
from mm2chain import params, synth
mm2chain.init()
for profile in ("mixed", "dense", "colinear"):
    off1, a1 = synth.make_stream(profile, 4096, 5000, seed=3, device="cuda")
    off, a = synth.replicate(off1, a1.cpu(), 8)
    a = a.cuda()
    f = torch.empty(a.shape[0], dtype=torch.int32, device="cuda"); p = torch.empty_like(f)
    res = {}
    for name, (ms, mi) in (("max_skip INT_MAX (C++ loop)", (2**31 - 1, 1024)), ("max_skip 1023 (hand-written loop)", (1023, 1024))):
        P = params.make_params(5000, 5000, 500, ms, mi, 1.0, 0, 1, 15, mm2chain.MM2C_F_IGNORE_SEG)
        plan = mm2chain.ChainPlan(P, off.numpy())
        for _ in range(3):
            plan.run(a, f, p)
        torch.cuda.synchronize()
        res[name] = (plan.last_kernel_ms(), f.clone(), p.clone(), plan.last_variant())
        plan.close()
    (k0, r0), (k1, r1) = res.items()
    same = bool(torch.equal(r0[1], r1[1]) and torch.equal(r0[2], r1[2]))
    print(f"{profile}: {k0}: {r0[0]:.2f} ms [{r0[3]}]\n          {k1}: {r1[0]:.2f} ms [{r1[3]}]; identical f / p: {same}")
mm2chain.shutdown()
