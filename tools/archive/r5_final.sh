#!/bin/bash
# round 5, evidence run: PMC profiles of the four bench shapes the round reports, the throughput table and the driver's line
set -u
O=gpurun_out/r5_final; mkdir -p $O
timeout -k 10 600 bash tools/profile.sh r5_mixed --profile mixed && echo prof mixed done
timeout -k 10 600 bash tools/profile.sh r5_dense --profile dense && echo prof dense done
timeout -k 10 600 bash tools/profile.sh r5_ava_ont_mixed --preset ava-ont --profile mixed && echo prof ava done
timeout -k 10 600 bash tools/profile.sh r5_asm20_mixed --preset asm20 --profile mixed && echo prof asm20 done
timeout -k 10 900 bash tools/results_table.sh > $O/results_table.log 2>&1; cp gpurun_out/results_table.md $O/ 2>/dev/null
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 800 $O/bench_default.json
