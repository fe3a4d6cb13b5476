#!/bin/bash
# prepass (chain_window_start) and DP kernel ms of a few streams (GPU box); arguments: library variants (default: the in-tree library)
for A in "--profile mixed" "--profile dense" "--preset asm20 --profile mixed" "--preset ava-ont --profile mixed" "--profile mixed --ragged"; do
  for NAME in ${@:-base}; do
    LIB=$PWD/minimap2-fpga_amd/variants/$NAME.so; [ "$NAME" = base ] && LIB=$PWD/minimap2-fpga_amd/libmm2chain_hip.so
    MM2C_LIB_PATH=$LIB timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary --steps 3 --warmup 1 $A 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$NAME [$A] prepass %.3f ms  DP %.2f ms  step %.2f ms' % (r['prepass_kernel_ms_avg'], r['kernel_ms_avg'], d['ms_per_step']))"
  done
done
