#!/bin/bash
# kernel-iteration round on the GPU box: parity tests, then the DP kernel time of the streams the round's targets name
# usage: tools/iter2.sh <tag> [tests=1]
TAG=$1; TESTS=${2:-1}
if [ "$TESTS" = "1" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/iter_$TAG.tests.log 2>&1
  tail -2 gpurun_out/iter_$TAG.tests.log
fi
run() {
  timeout -k 10 300 python bench.py --cpu-seconds 0 --no-secondary --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "
import sys,json
l=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(l[-1]); print('$TAG %-28s kernel %.2f ms  verified %s' % ('$*', d['roofline']['kernel_ms_avg'], d['verified_vs_oracle']))
except Exception as e: print('$TAG $* FAILED', l[-2:])"
}
run --profile mixed
run --profile dense
run --profile colinear
run --profile mixed --ragged
run --preset asm20 --profile mixed
run --preset ava-ont --profile mixed
run --preset ava-ont --profile colinear
