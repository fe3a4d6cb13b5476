#!/usr/bin/env python3
"""round 5: mm2c_chain_batch_host from page-locked memory (bench.py's host_streamed_pinned: 8 192 reads x 5 000 anchors), with the last chunks of the pipeline run
with several waves per piece (mm2c_tune("pipe_coop_chunks", n)) and different pipeline chunkings"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mm2chain
from mm2chain import params, synth
mm2chain.init(0)
P = params.map_ont()
off, a = synth.make_stream("mixed", 8192, 5000, seed=20240, device="cuda")
a_host = a.cpu().numpy().view(np.uint64); off_host = off.numpy()
pa = mm2chain.PinnedArray(a_host.shape, np.uint64); pf = mm2chain.PinnedArray((a_host.shape[0],), np.int32); pp = mm2chain.PinnedArray((a_host.shape[0],), np.int32)
pa.array[:] = a_host
mm2chain.chain_batch_host_into(P, off_host, pa.array, pf.array, pp.array)
f0, p0 = pf.array.copy(), pp.array.copy()
for pieces in (8, 12, 16):
    mm2chain.tune("pipeline_pieces", pieces) if False else None
for coop in (0, 1, 2, 3, 4, 8):
    mm2chain.tune("pipe_coop_chunks", coop)
    ts = []
    for _ in range(12):
        t = time.perf_counter(); mm2chain.chain_batch_host_into(P, off_host, pa.array, pf.array, pp.array); ts.append(time.perf_counter() - t)
    ok = np.array_equal(pf.array, f0) and np.array_equal(pp.array, p0)
    print(f"pipe_coop_chunks={coop}: best {min(ts)*1e3:.2f} ms = {a_host.shape[0] / min(ts) / 1e9:.2f} G anchors/s, median {sorted(ts)[6]*1e3:.2f} ms, same results {ok}; last launch: {mm2chain.last_host_variant()}", flush=True)
mm2chain.tune("pipe_coop_chunks", 0)
