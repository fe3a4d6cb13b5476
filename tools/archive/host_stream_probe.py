#!/usr/bin/env python3
"""PCIe-inclusive rate of mm2c_chain_batch_host on a full-size batch (65 536 reads x 5 000 anchors), pageable vs page-locked,
with and without the chunked two-stream pipeline (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import torch, mm2chain
from mm2chain import params, synth
mm2chain.init()
P = params.map_ont()
off1, a1 = synth.make_stream("mixed", 8192, 5000, seed=20240, device="cuda")
off, a = synth.replicate(off1, a1, 8)
off = off.numpy(); total = int(off[-1])
pa = mm2chain.PinnedArray((total, 2), np.uint64); pf = mm2chain.PinnedArray((total,), np.int32); pp = mm2chain.PinnedArray((total,), np.int32)
pa.array[:] = a.cpu().numpy().view(np.uint64)
for name, chunk in (("single pass", 1 << 30), ("pipelined 40 Mi chunks", 40 << 20), ("pipelined 20 Mi chunks", 20 << 20)):
    mm2chain.tune("pipeline_chunk_anchors", chunk)
    mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array)
    t0 = time.perf_counter(); mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array); dt = time.perf_counter() - t0
    print(f"page-locked, {name}: {dt*1e3:.1f} ms -> {total/dt/1e9:.2f} G anchors/s")
