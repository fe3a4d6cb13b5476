#!/usr/bin/env python3
"""whole mm_chain_dp for a batch: GPU DP + epilogue on the GPU vs on host threads.
usage: python tools/epilogue_probe.py [n_reads] [anchors_per_read] [profile] [--device-only]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minimap2-fpga_amd"))
import ctypes as C
import numpy as np
import torch
import mm2chain
from mm2chain import params, synth, _native as N

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_reads = int(args[0]) if len(args) > 0 else 2000
per = int(args[1]) if len(args) > 1 else 10000
profile = args[2] if len(args) > 2 else "mixed"
device_only = "--device-only" in sys.argv
pinned_only = "--pinned-only" in sys.argv
mm2chain.init()
P = params.map_ont()
distinct = min(n_reads, 4096)                       # as bench.py: distinct reads tiled up to the batch size
off_t, a_t = synth.make_stream(profile, distinct, (per, per), seed=5)
if n_reads > distinct:
    off_t, a_t = synth.replicate(off_t, a_t, n_reads // distinct)
    n_reads = off_t.numel() - 1
total = int(off_t[-1])

# HBM-resident: plan.run + plan.chains
d_a = a_t.cuda(); d_f = torch.empty(total, dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
plan = mm2chain.ChainPlan(P, off_t.numpy())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.run(d_a, d_f, d_p)
    u_off, u, b_off, b = plan.chains(d_a, d_f, d_p, 3, 40)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
n_u, n_b = int(u_off[-1]), int(b_off[-1])
print(f"{profile}: anchors {total}  chains {n_u}  chained anchors {n_b}")
print(f"HBM-resident: DP {plan.last_kernel_ms():.2f} ms (+ prepass {plan.last_prepass_ms():.2f})  epilogue {plan.last_epilogue_ms():.2f} ms  "
      f"wall {wall*1e3:.2f} ms = {total/wall/1e9:.3f} G anchors/s for the whole mm_chain_dp")
plan.close(); del d_a, d_f, d_p, u, b
if not device_only:
    lib = N.load()
    off = off_t.numpy(); a = a_t.numpy().view(np.uint64)
    u_off = np.zeros(n_reads + 1, np.int64); b_off = np.zeros(n_reads + 1, np.int64)
    u = np.zeros(total, np.uint64); b = np.zeros((total, 2), np.uint64)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    for nt in (() if pinned_only else (0, 1, 4, 16)):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            rc = lib.mm2c_mm_chain_dp_batch_host(C.byref(P), 3, 40, n_reads, ptr(off), ptr(a), nt, ptr(u_off), ptr(u), ptr(b_off), ptr(b))
            best = min(best, time.perf_counter() - t0)
        assert rc == 0 and u_off[-1] == n_u and b_off[-1] == n_b
        print(f"host buffers, epilogue {'on the GPU' if nt == 0 else 'on %2d host threads' % nt}: {best*1e3:.1f} ms  {total/best/1e9:.3f} G anchors/s")
    # page-locked caller buffers (mm2c_pinned_alloc): the pipelined path at PCIe rate
    pa = mm2chain.PinnedArray(a.shape, np.uint64); pa.array[:] = a
    pu = mm2chain.PinnedArray(u.shape, np.uint64); pb = mm2chain.PinnedArray(b.shape, np.uint64)
    for chunk in (10 << 20, 20 << 20, 40 << 20, 80 << 20):
        if chunk * 2 > total and chunk != 20 << 20:
            continue
        mm2chain.tune("pipeline_chunk_anchors", chunk)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = lib.mm2c_mm_chain_dp_batch_host(C.byref(P), 3, 40, n_reads, ptr(off), ptr(pa.array), 0, ptr(u_off), ptr(pu.array), ptr(b_off), ptr(pb.array))
            best = min(best, time.perf_counter() - t0)
        assert rc == 0 and u_off[-1] == n_u
        print(f"page-locked host buffers, epilogue on the GPU, chunks of {chunk >> 20} Mi anchors: {best*1e3:.1f} ms  {total/best/1e9:.3f} G anchors/s")
    mm2chain.tune("pipeline_chunk_anchors", 20 << 20)
mm2chain.shutdown()
