#!/bin/bash
# DP kernel time against the bar of the ring-size classes (tenths of an expected far tile per anchor; MM2C_FAR_RING_THRESHOLD) -- GPU box
for THR in ${@:-7 8 9 10 12}; do
  for ARGS in "--profile mixed --ragged" "--profile dense" "--profile dense --ragged" "--preset asm20 --ragged" "--preset ava-ont" "--preset asm20"; do
    MM2C_FAR_RING_THRESHOLD=$THR timeout -k 10 200 python bench.py --cpu-seconds 0 --no-secondary --steps 3 --warmup 1 $ARGS 2>/dev/null | python3 -c "
import sys,json
l=sys.stdin.read().strip().splitlines()
d=json.loads(l[-1]); print('thr=$THR $ARGS: kernel %.2f ms  verified %s' % (d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))"
  done
done
