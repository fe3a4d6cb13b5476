#!/bin/bash
# what clock does the GPU run lone calls at?  samples rocm-smi while a loop of lone run_chaining_on_hw calls runs, then the same loop beside a busy GPU (bench.py in a loop)
MM2C_QUIET=1 python - <<'PY' &
import os, sys, time, numpy as np
sys.path.insert(0, "minimap2-fpga_amd")
import mm2chain
from mm2chain import params, synth
mm2chain.init()
off, a = synth.make_stream("mixed", 1, 5000, seed=1); t = a.numpy().view(np.uint64)
for rnd in range(12):
    ts = []
    t_end = time.time() + 1.0
    while time.time() < t_end:
        t0 = time.perf_counter(); mm2chain.run_chaining_on_hw(5000, 5000, 5000, 500, 15, 0.15, t); ts.append(time.perf_counter() - t0)
    print(f"round {rnd}: {len(ts)} calls, best {min(ts)*1e3:.3f} ms median {sorted(ts)[len(ts)//2]*1e3:.3f} ms", flush=True)
PY
P1=$!
sleep 4
for k in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|fclk"; sleep 1; done
echo "--- now with the GPU busy beside it"
timeout -k 5 8 python bench.py --steps 60 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1 &
sleep 5
rocm-smi --showclocks 2>/dev/null | grep -E "sclk"
wait $P1
