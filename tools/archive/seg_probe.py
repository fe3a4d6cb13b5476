#!/usr/bin/env python3
"""Does cutting tasks at empty windows help BATCH throughput on real-like anchor lists?  (GPU box)"""
import os, struct, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, mm2chain
from mm2chain import params
W = "/tmp/segprobe"; os.makedirs(W, exist_ok=True)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/make_synth_genome.py"), W + "/syn", "--genome-mb", "50", "--reads", "2000"], stdout=subprocess.DEVNULL)
dump = W + "/dump.bin"
if os.path.exists(dump): os.unlink(dump)
subprocess.check_call([os.path.join(ROOT, "oracle/_ref/mm2_refhost"), "-t", "1", W + "/syn.ref.fa", W + "/syn.reads.fa"], env=dict(os.environ, MM2O_DUMP=dump), stdout=subprocess.DEVNULL)
raw = open(dump, "rb").read(); pos = 0; tasks = []
while pos < len(raw):
    n, = struct.unpack_from("<q", raw, pos); pos += 8 + 40
    tasks.append(np.frombuffer(raw, dtype=np.uint64, count=2 * n, offset=pos).reshape(n, 2).copy()); pos += 16 * n
reps = 10
a = np.concatenate(tasks * reps)
off = np.concatenate(([0], np.cumsum([len(t) for t in tasks] * reps))).astype(np.int64)
print(f"{len(off)-1} tasks, {a.shape[0]} anchors, mean {a.shape[0]/(len(off)-1):.0f} per task, max {max(len(t) for t in tasks)}")
mm2chain.init()
P = params.map_ont()
# device-resident plan (no cutting): kernel time only
d_a = torch.from_numpy(a.view(np.int64)).cuda(); d_f = torch.empty(a.shape[0], dtype=torch.int32, device="cuda"); d_p = torch.empty_like(d_f)
pl = mm2chain.ChainPlan(P, off)
for _ in range(3): pl.run(d_a, d_f, d_p)
torch.cuda.synchronize()
print(f"plan (whole tasks): DP kernel {pl.last_kernel_ms():.2f} ms, prepass {pl.last_prepass_ms():.2f} ms -> {a.shape[0]/pl.last_kernel_ms()/1e6:.2f} G anchors/s")
ref_f = d_f.cpu().numpy()
for seg_min in (0, 256, 1024):
    mm2chain.tune("seg_min", seg_min)
    mm2chain.chain_batch_host(P, off, a)
    t0 = time.perf_counter(); f, p = mm2chain.chain_batch_host(P, off, a); dt = time.perf_counter() - t0
    print(f"host batch seg_min {seg_min}: {dt*1e3:.1f} ms total (PCIe included), same as plan: {bool((f == ref_f).all())}")
