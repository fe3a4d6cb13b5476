#!/usr/bin/env python3
"""Per-call latency of the synchronous host path on real-like anchor lists (GPU box).  Generates a synthetic genome, maps reads
with the reference host objects (CPU chaining, MM2O_DUMP) to capture the anchor lists that reach mm_chain_dp, then times
mm2c_chain_task_host per task and the kernels alone (plan, device resident)."""
import os, struct, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, mm2chain
from mm2chain import params
W = "/tmp/latprobe"; os.makedirs(W, exist_ok=True)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/make_synth_genome.py"), W + "/syn", "--genome-mb", "50", "--reads", "400"], stdout=subprocess.DEVNULL)
dump = W + "/dump.bin"
if os.path.exists(dump): os.unlink(dump)
subprocess.check_call([os.path.join(ROOT, "oracle/_ref/mm2_refhost"), W + "/syn.ref.fa", W + "/syn.reads.fa"], env=dict(os.environ, MM2O_DUMP=dump), stdout=subprocess.DEVNULL)
raw = open(dump, "rb").read(); pos = 0; tasks = []
while pos < len(raw):
    n, = struct.unpack_from("<q", raw, pos); pos += 8 + 40
    tasks.append(np.frombuffer(raw, dtype=np.uint64, count=2 * n, offset=pos).reshape(n, 2).copy()); pos += 16 * n
mm2chain.init()
P = params.map_ont()
import oracle_binding as ob
for seg_min in (0, 64, 256):
    mm2chain.tune("seg_min", seg_min)
    for t in tasks[:20]: mm2chain.chain_task(P, t, ob.avg_qspan(t))
    t0 = time.perf_counter()
    for t in tasks: mm2chain.chain_task(P, t, 0.15)
    dt = time.perf_counter() - t0
    print(f"seg_min {seg_min}: {len(tasks)} calls, {sum(len(t) for t in tasks)} anchors, {dt/len(tasks)*1e6:.1f} us per call")
tiny = tasks[0][:8]
t0 = time.perf_counter()
for _ in range(400): mm2chain.chain_task(P, tiny, 0.15)
print(f"8-anchor task: {(time.perf_counter()-t0)/400*1e6:.1f} us per call (fixed overhead)")
km, pm, ns = [], [], []
for t in tasks[:200]:
    d_a = torch.from_numpy(t.view(np.int64)).cuda(); d_f = torch.empty(len(t), dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
    pl = mm2chain.ChainPlan(P, [0, len(t)])
    pl.run(d_a, d_f, d_p); pl.run(d_a, d_f, d_p)
    km.append(pl.last_kernel_ms()); pm.append(pl.last_prepass_ms()); ns.append(len(t)); pl.close()
km, pm, ns = np.array(km), np.array(pm), np.array(ns)
print(f"whole task as one wave: DP kernel mean {km.mean()*1e3:.1f} us (n mean {ns.mean():.0f}) = {km.sum()/ns.sum()*1e6:.1f} ns/anchor, prepass mean {pm.mean()*1e3:.1f} us")
t0 = time.perf_counter()
for t in tasks: ob.chain_fpv(P, t, 0.15)
print(f"CPU oracle (1 thread, via ctypes): {(time.perf_counter()-t0)/len(tasks)*1e6:.1f} us per task")
