#!/bin/bash
# round 5: a variant build of the DP kernel (variants/<name>.so, tools/probe_build.sh) against the shipped library on the bench shapes, alternating, same box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
V=$1; shift
for CFG in "" "--profile dense" "--preset asm20" "--preset ava-ont"; do
  for REP in 1 2; do
    for LIB in "" "$REPO/minimap2-fpga_amd/variants/$V.so"; do
      R=$(MM2C_LIB_PATH=$LIB timeout -k 10 300 python3 bench.py $CFG --steps 4 --warmup 1 --cpu-seconds 0 --no-secondary --no-e2e 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.2f ms kernel, verified %s' % (d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))")
      echo "[$CFG] lib=${LIB##*/}: $R"
    done
  done
done
