#!/bin/bash
# round 5: the long ring in its q24 form against its 32-bit form (MM2C_Q24_RING=0) on the streams whose tasks take the long ring
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for CFG in "--preset ava-ont" "--preset ava-ont --profile colinear" "--profile dense --ragged" "--preset asm20 --ragged" "--ragged" ""; do
  for Q in 1 0; do
    R=$(MM2C_Q24_RING=$Q timeout -k 10 300 python3 bench.py $CFG --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary --no-e2e 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.2f ms kernel, %.2f ms step, verified %s' % (d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['verified_vs_oracle']))")
    echo "[$CFG] q24=$Q: $R"
  done
done
