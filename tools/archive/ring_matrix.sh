#!/bin/bash
# DP kernel time of the streams whose windows leave the short LDS ring, with the ring-size classes off / per task / forced (GPU box)
for FR in 0 1 2; do
  for ARGS in "--profile dense" "--preset asm20" "--preset ava-ont" "--profile dense --preset asm20" "--profile mixed"; do
    MM2C_FAR_RING=$FR timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary --steps 3 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import sys,json
l=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(l[-1]); print('far_ring=$FR $ARGS: kernel %.2f ms  verified %s' % (d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))
except Exception as e: print('far_ring=$FR $ARGS FAILED', l[-2:])"
  done
done
