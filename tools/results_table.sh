#!/bin/bash
# the throughput table of profiles/rN_results.md: bench.py over the presets / stream profiles, one row each (runs on the GPU box)
OUT=gpurun_out/results_table.md
echo "| preset | stream profile | reads x anchors | G anchors/s | DP kernel ms / step | verified |" > $OUT
echo "|---|---|---|---|---|---|" >> $OUT
row() {   # label-preset, label-profile, bench args...
  local lp="$1" lf="$2"; shift; shift
  timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 3 --warmup 1 --no-secondary "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print('| $lp | $lf | %d x %d | %.2f | %.1f | %s |' % (c['reads_per_gpu_per_step'], c['anchors_per_read'], d['value'] / 1e9, d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))" >> $OUT || echo "| $lp | $lf | FAILED | | | |" >> $OUT
}
row "map-ont" "mixed (headline)" --profile mixed
row "map-ont" "dense" --profile dense
row "map-ont" "sparse" --profile sparse
row "map-ont" "colinear" --profile colinear
row "map-ont" "mixed, ragged (500..9500 anchors per read)" --profile mixed --ragged
row "asm20" "mixed" --preset asm20 --profile mixed --anchors-per-read 7500
row "asm20" "colinear" --preset asm20 --profile colinear --anchors-per-read 7500
row "ava-ont" "mixed" --preset ava-ont --profile mixed --reads 16384 --anchors-per-read 20000
row "ava-ont" "colinear" --preset ava-ont --profile colinear --reads 16384 --anchors-per-read 20000
row "map-ont" "dense, ragged" --profile dense --ragged
row "asm20" "mixed, ragged" --preset asm20 --profile mixed --ragged
row "map-ont, general variant" "mixed" --profile mixed --general
row "map-ont, gap_scale 0.8" "mixed" --profile mixed --gap-scale 0.8
cat $OUT
