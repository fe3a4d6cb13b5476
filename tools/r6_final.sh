#!/bin/bash
# round 6, evidence run: PMC profiles of the bench shapes (the DP loop of the headline kernel is round 5's; profiles/traffic.json is keyed by the kernel sources' hash), the
# long-read profile (kernel stats + PMC of the cooperative kernel), the throughput table and the driver's line
set -u
O=gpurun_out/r6_final; mkdir -p $O
step=${1:-all}
if [ $step = all ] || [ $step = prof ]; then
timeout -k 10 600 bash tools/profile.sh r6_mixed --profile mixed --no-long && echo prof mixed done
timeout -k 10 600 bash tools/profile.sh r6_dense --profile dense --no-long && echo prof dense done
timeout -k 10 600 bash tools/profile.sh r6_ava_ont_mixed --preset ava-ont --profile mixed && echo prof ava done
timeout -k 10 600 bash tools/profile.sh r6_asm20_mixed --preset asm20 --profile mixed && echo prof asm20 done
fi
if [ $step = all ] || [ $step = long ]; then
LONG_PROFILE=1 LONG_PMC="--routes coop16 --sizes 256x1000000 --distinct 2" timeout -k 10 900 tools/long_reads.sh r6_long > $O/long_reads.log 2>&1; echo long reads done
fi
if [ $step = all ] || [ $step = bench ]; then
timeout -k 10 900 bash tools/results_table.sh > $O/results_table.log 2>&1; cp gpurun_out/results_table.md $O/ 2>/dev/null
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
fi
