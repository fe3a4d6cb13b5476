#!/usr/bin/env python3
"""host-streamed batch (mm2c_chain_batch_host from page-locked memory: the bench's host_streamed_pinned figure) under chunk schedules: pipeline_taper 0 .. 4"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch, mm2chain
from mm2chain import params, synth
mm2chain.init()
P = params.map_ont()
off, a = synth.make_stream("mixed", 8192, 5000, seed=20240, device="cuda")
a_h = a.cpu().numpy().view(np.uint64); off_h = off.numpy()
pa = mm2chain.PinnedArray(a_h.shape, np.uint64); pf = mm2chain.PinnedArray((a_h.shape[0],), np.int32); pp = mm2chain.PinnedArray((a_h.shape[0],), np.int32)
pa.array[:] = a_h
ref = None
for taper in (0, 1, 2, 3, 4, 3, 0):
    mm2chain.tune("pipeline_taper", taper)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); mm2chain.chain_batch_host_into(P, off_h, pa.array, pf.array, pp.array); ts.append(time.perf_counter() - t0)
    if ref is None: ref = (pf.array.copy(), pp.array.copy())
    same = bool(np.array_equal(pf.array, ref[0]) and np.array_equal(pp.array, ref[1]))
    print(f"pipeline_taper {taper}: best {min(ts)*1e3:.2f} ms = {a_h.shape[0]/min(ts)/1e9:.3f} G anchors/s, median {np.median(ts)*1e3:.2f} ms; same results: {same}", flush=True)
mm2chain.shutdown()
