#!/usr/bin/env python3
"""one configuration of tools/host_stream_probe2.py, three timed calls -- for a rocprofv3 kernel + memory-copy trace (GPU box).
usage: host_stream_trace.py <pieces> [reads]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import torch, mm2chain
from mm2chain import params, synth
mm2chain.init()
P = params.map_ont()
pieces = int(sys.argv[1]); reads = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
off, a = synth.make_stream("mixed", reads, 5000, seed=20240, device="cuda")
off = off.numpy(); total = int(off[-1])
pa = mm2chain.PinnedArray((total, 2), np.uint64); pf = mm2chain.PinnedArray((total,), np.int32); pp = mm2chain.PinnedArray((total,), np.int32)
pa.array[:] = a.cpu().numpy().view(np.uint64)
mm2chain.tune("pipeline_pieces", pieces); mm2chain.tune("pipeline_min_chunk", 1 << 20)
for _ in range(4):
    t0 = time.perf_counter(); mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array); dt = time.perf_counter() - t0
    print(f"{pieces} pieces: {dt*1e3:.2f} ms -> {total/dt/1e9:.2f} G anchors/s", flush=True)
    time.sleep(0.01)
