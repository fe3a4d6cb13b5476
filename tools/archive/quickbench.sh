#!/bin/bash
# usage: tools/quickbench.sh <tag> "<extra bench args>" [profiles...]   -- runs on the GPU box
TAG=$1; EXTRA=$2; shift; shift
PROFS=${@:-mixed dense sparse colinear}
for prof in $PROFS; do
  timeout -k 10 300 python bench.py --profile $prof --cpu-seconds 0 --steps 3 --warmup 1 $EXTRA 2>&1 | python3 -c "
import sys,json
lines=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(lines[-1]); print('$TAG $prof %.3f Ganchors/s kernel %.2f ms verified %s' % (d['value']/1e9, d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))
except Exception as e:
    print('$TAG $prof FAILED', lines[-3:])
" >> gpurun_out/quick.log
done
