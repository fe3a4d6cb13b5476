#!/bin/bash
# On the GPU box: bench.py (DP only) for each library variant of minimap2-fpga_amd/variants/.   usage: tools/probe_run.sh "<bench args>" name [name ...]
ARGS=$1; shift
for NAME in "$@"; do
  LIB=$PWD/minimap2-fpga_amd/variants/$NAME.so
  [ "$NAME" = base ] && LIB=$PWD/minimap2-fpga_amd/libmm2chain_hip.so
  MM2C_LIB_PATH=$LIB timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary --steps 3 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import sys,json
l=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(l[-1]); print('$NAME: kernel %.2f ms  verified %s' % (d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))
except Exception as e: print('$NAME FAILED', l[-2:])"
done
