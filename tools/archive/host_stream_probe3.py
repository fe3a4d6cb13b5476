#!/usr/bin/env python3
"""does a call from pageable memory slow down later calls from page-locked memory?  (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd"))
import torch, mm2chain
from mm2chain import params, synth
mm2chain.init()
P = params.map_ont()
off, a = synth.make_stream("mixed", 8192, 5000, seed=20240, device="cuda")
off = off.numpy(); total = int(off[-1])
a_host = a.cpu().numpy().view(np.uint64)
pa = mm2chain.PinnedArray((total, 2), np.uint64); pf = mm2chain.PinnedArray((total,), np.int32); pp = mm2chain.PinnedArray((total,), np.int32)
pa.array[:] = a_host
def timed(label, fn):
    fn(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best*1e3:.2f} ms -> {total/best/1e9:.2f} G anchors/s", flush=True)
timed("page-locked, first", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
timed("pageable", lambda: mm2chain.chain_batch_host(P, off, a_host))
timed("page-locked, after pageable", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
sp_streams = [torch.cuda.Stream(priority=p) for p in (0, -1, 0, -1)]
timed("page-locked, after creating 4 more streams", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
x = torch.empty(3 << 30, dtype=torch.uint8, device="cuda"); x.zero_(); torch.cuda.synchronize()
timed("page-locked, after a 3 GB torch allocation", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
del x
off1, a1 = synth.make_stream("mixed", 8192, 5000, seed=20240, device="cuda")
offb, ab = synth.replicate(off1, a1, 8)
fb = torch.empty(int(offb[-1]), dtype=torch.int32, device="cuda"); pb = torch.empty_like(fb)
plan = mm2chain.ChainPlan(P, offb.numpy()); plan.run(ab, fb, pb); torch.cuda.synchronize()
timed("page-locked, after a 65 536-read plan ran", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
r = plan.chains(ab, fb, pb, 3, 40); torch.cuda.synchronize(); del r
timed("page-locked, after the device epilogue ran", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
m_k, h_k = synth.matches_from_anchors(a_host[:5000], 1 << 20)
sp = mm2chain.SeedPlan(np.array([0, m_k.size], np.int64), np.array([0, h_k.size], np.int64))
d_as = sp.run(torch.from_numpy(m_k.view(np.uint8)).cuda(), torch.from_numpy(h_k.view(np.int64)).cuda(), torch.full((1,), 1 << 20, dtype=torch.int32, device="cuda")); torch.cuda.synchronize()
timed("page-locked, after a seed plan ran", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
sp.close()
timed("page-locked, after the seed plan was closed", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
# the seed-hit figure of bench.py: 256 reads x 32 replicas through every size class of the tie replay (all helper streams)
n_s = 256; qlen_s = 1 << 20
ms_, hs_, mo_, ao_ = [], [], [0], [0]
for k in range(n_s):
    m_k, h_k = synth.matches_from_anchors(a_host[int(off[k]):int(off[k + 1])], qlen_s)
    m_k["cr_off"] += ao_[-1]
    ms_.append(m_k); hs_.append(h_k); mo_.append(mo_[-1] + m_k.size); ao_.append(ao_[-1] + h_k.size)
rep = 32
m1_, h1_ = np.concatenate(ms_), np.concatenate(hs_)
mt_ = np.tile(m1_, rep); mt_["cr_off"] += np.repeat(np.arange(rep, dtype=np.int64) * h1_.size, m1_.size)
mo_t = np.concatenate([[0], np.tile(np.diff(mo_), rep).cumsum()]).astype(np.int64)
ao_t = np.concatenate([[0], np.tile(np.diff(ao_), rep).cumsum()]).astype(np.int64)
sp = mm2chain.SeedPlan(mo_t, ao_t)
d_m = torch.from_numpy(mt_.view(np.uint8)).cuda(); d_h = torch.from_numpy(np.tile(h1_, rep).view(np.int64)).cuda()
d_q = torch.full((n_s * rep,), qlen_s, dtype=torch.int32, device="cuda")
d_as = sp.run(d_m, d_h, d_q); torch.cuda.synchronize()
timed("page-locked, after the 8 192-read seed plan of bench.py ran", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
sp.close(); del d_m, d_h, d_as
timed("page-locked, after it was closed", lambda: mm2chain.chain_batch_host_into(P, off, pa.array, pf.array, pp.array))
pa2 = mm2chain.PinnedArray((total, 2), np.uint64); pf2 = mm2chain.PinnedArray((total,), np.int32); pp2 = mm2chain.PinnedArray((total,), np.int32)
pa2.array[:] = a_host
timed("page-locked, freshly allocated buffers", lambda: mm2chain.chain_batch_host_into(P, off, pa2.array, pf2.array, pp2.array))
