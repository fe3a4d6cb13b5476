#!/usr/bin/env python3
"""Re-fit of the reference's HW/SW split model for MI355X (reference: hw_sw_split/find_params.py fits the same two
linear models from `param n total_subparts total_trip_count hw_ms sw_ms` lines printed at chain.c:333).
   hw_ms = K1_HW * n + K2_HW * total_subparts + C_HW          (chain.c:80)
   sw_ms = K_SW * total_trip_count + C_SW                      (chain.c:81)
Here: hw_ms = one synchronous mm2c_chain_task_host call (PCIe + launch + DP, pieces cut at empty windows), sw_ms = the CPU oracle
(port of chain.c:184-238) on one host core; tasks = anchor lists of simulated ONT reads on a synthetic genome, captured from the
reference host objects, plus synthetic dense tasks to spread the regressors.  Prints constants in the form of chain_hardware.h:19-23."""
import os, struct, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "minimap2-fpga_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mm2chain
from mm2chain import params, synth
import oracle_binding as ob
W = "/tmp/splitfit"; os.makedirs(W, exist_ok=True)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/make_synth_genome.py"), W + "/syn", "--genome-mb", "50", "--reads", "600"], stdout=subprocess.DEVNULL)
dump = W + "/dump.bin"
if os.path.exists(dump): os.unlink(dump)
subprocess.check_call([os.path.join(ROOT, "oracle/_ref/mm2_refhost"), W + "/syn.ref.fa", W + "/syn.reads.fa"], env=dict(os.environ, MM2O_DUMP=dump), stdout=subprocess.DEVNULL)
raw = open(dump, "rb").read(); pos = 0; tasks = []
while pos < len(raw):
    n, = struct.unpack_from("<q", raw, pos); pos += 8 + 40
    tasks.append(np.frombuffer(raw, dtype=np.uint64, count=2 * n, offset=pos).reshape(n, 2).copy()); pos += 16 * n
rng = np.random.default_rng(1)
for prof in ("mixed", "dense", "colinear"):
    for n in rng.integers(200, 8000, 60):
        tasks.append(synth.make_stream(prof, 1, int(n), seed=int(rng.integers(1 << 30)))[1].numpy().view(np.uint64))
mm2chain.init()
P = params.map_ont()
rows = []
for t in tasks[:20]:
    mm2chain.chain_task(P, t, 0.15)
for t in tasks:
    n = t.shape[0]
    _, tot_sub, tot_trip = ob.predict(t, P.max_dist_x)
    h = []
    for _ in range(3):
        t0 = time.perf_counter(); mm2chain.chain_task(P, t, 0.15); h.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ob.chain_fpv(P, t, 0.15); sw = time.perf_counter() - t0
    rows.append((n, tot_sub, tot_trip, min(h) * 1e3, sw * 1e3))
R = np.array(rows, dtype=np.float64)
A = np.stack((R[:, 0], R[:, 1], np.ones(len(R))), 1)
(k1, k2, c_hw), *_ = np.linalg.lstsq(A, R[:, 3], rcond=None)
B = np.stack((R[:, 2], np.ones(len(R))), 1)
(k_sw, c_sw), *_ = np.linalg.lstsq(B, R[:, 4], rcond=None)
r2 = lambda y, yh: 1 - ((y - yh) ** 2).sum() / ((y - y.mean()) ** 2).sum()
print(f"tasks {len(R)}, n {R[:,0].min():.0f}..{R[:,0].max():.0f}, hw_ms {R[:,3].min():.3f}..{R[:,3].max():.3f}, sw_ms {R[:,4].min():.3f}..{R[:,4].max():.3f}")
print(f"#define MI355X_ONT_K1_HW {k1:.10g}\n#define MI355X_ONT_K2_HW {k2:.10g}\n#define MI355X_ONT_C_HW {c_hw:.10g}   // R^2 {r2(R[:,3], A @ [k1,k2,c_hw]):.3f}")
print(f"#define MI355X_ONT_K_SW {k_sw:.10g}\n#define MI355X_ONT_C_SW {c_sw:.10g}   // R^2 {r2(R[:,4], B @ [k_sw,c_sw]):.3f}")
gpu_wins = (A @ [k1, k2, c_hw]) < (B @ [k_sw, c_sw])
print(f"model sends {gpu_wins.mean()*100:.0f} % of these single-call tasks to the GPU; measured GPU faster in {(R[:,3] < R[:,4]).mean()*100:.0f} %")
print("reference constants (VU9P, chain_hardware.h:19-23): K1_HW 2.992e-4, K2_HW 1.215e-5, C_HW 0.319, K_SW 5.234e-6, C_SW -1.0015")
