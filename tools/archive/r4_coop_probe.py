#!/usr/bin/env python3
"""timing of one run_chaining_on_hw call (V2 scalars, lone task) on the library named by MM2C_LIB_PATH; results are NOT checked (probe builds switch work off)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import mm2chain
from mm2chain import params, synth
mm2chain.init()
out = []
for prof in ("mixed", "colinear"):
    off, a = synth.make_stream(prof, 1, 5000, seed=1)
    t = a.numpy().view(np.uint64)
    P = params.map_ont()
    ts = []
    for k in range(40):
        t0 = time.perf_counter(); mm2chain.run_chaining_on_hw(5000, 5000, 5000, 500, 15, 0.15, t); ts.append(time.perf_counter() - t0)
    tv = []
    for k in range(40):
        t0 = time.perf_counter(); mm2chain.chain_task(P, t, 0.15); tv.append(time.perf_counter() - t0)
    out.append(f"{prof}: V2 {min(ts[5:])*1e3:.3f} ms, V1 {min(tv[5:])*1e3:.3f} ms")
print(os.path.basename(os.environ.get("MM2C_LIB_PATH", "base")), "; ".join(out))
