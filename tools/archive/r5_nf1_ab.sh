#!/bin/bash
# round 5: f / p ring of the long (q24) ring: 2 tiles (shipped) against 4 (variants/nf1_4.so, MM2C_LIB_PATH)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for CFG in "--preset ava-ont" "--profile dense --ragged" "--preset asm20 --ragged"; do
  for LIB in "" "$REPO/minimap2-fpga_amd/variants/nf1_4.so"; do
    R=$(MM2C_LIB_PATH=$LIB timeout -k 10 300 python3 bench.py $CFG --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary --no-e2e 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.2f ms kernel, %.2f ms step, verified %s' % (d['roofline']['kernel_ms_avg'], d['ms_per_step'], d['verified_vs_oracle']))")
    echo "[$CFG] lib=${LIB##*/}: $R"
  done
done
