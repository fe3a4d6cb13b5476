#!/usr/bin/env python3
"""PCIe-inclusive rate of mm2c_chain_batch_host on the sample bench.py reports (4096 reads x 5000 anchors, page-locked) as a function of how
the two-stream pipeline cuts it (GPU box).  usage: host_stream_probe2.py [reads]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import torch, mm2chain
from mm2chain import params, synth
mm2chain.init()
P = params.map_ont()
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
off, a = synth.make_stream("mixed", reads, 5000, seed=20240, device="cuda")
off = off.numpy(); total = int(off[-1])
pa = mm2chain.PinnedArray((total, 2), np.uint64); pf = mm2chain.PinnedArray((total,), np.int32); pp = mm2chain.PinnedArray((total,), np.int32)
pa.array[:] = a.cpu().numpy().view(np.uint64)
for pieces, min_chunk in ((1, 1 << 30), (2, 1 << 20), (4, 1 << 20), (8, 1 << 20), (16, 1 << 20), (5, 4 << 20)):
    mm2chain.tune("pipeline_pieces", pieces); mm2chain.tune("pipeline_min_chunk", min(min_chunk, 2**31 - 1))
    mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array); best = min(best, time.perf_counter() - t0)
    print(f"page-locked, {reads} reads, {pieces} pieces (min chunk {min_chunk}): {best*1e3:.2f} ms -> {total/best/1e9:.2f} G anchors/s", flush=True)
