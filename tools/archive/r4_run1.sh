#!/bin/bash
# round 4, first GPU call: what bounds the BASELINE config 4 / 5 stand-ins (profiles + an occupancy sweep) and what a lone call costs
set -u
O=gpurun_out/r4_run1; mkdir -p $O
b() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-secondary --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-60s kernel %.2f ms  prepass %.2f  verified %s' % ('$*', d['roofline']['kernel_ms_avg'], d.get('prepass_ms_avg', -1), d['verified_vs_oracle']))" ; }
{
for r in 16384 11520 7680 3840 1920 960; do b --preset ava-ont --profile mixed --reads $r; done
for r in 65536 16384 6656 3328 1664; do b --profile mixed --reads $r; done
} > $O/occupancy.txt 2>&1
cat $O/occupancy.txt
timeout -k 10 600 python tools/r4_lone_call.py > $O/lone_call.txt 2>&1; tail -20 $O/lone_call.txt
timeout -k 10 900 bash tools/profile.sh r4_ava_ont_mixed --preset ava-ont --profile mixed && echo prof1 done
timeout -k 10 900 bash tools/profile.sh r4_asm20_mixed --preset asm20 --profile mixed && echo prof2 done
