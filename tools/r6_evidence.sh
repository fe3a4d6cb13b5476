#!/bin/bash
# round 6, evidence after the two-width cooperative kernel: PMC profiles of the four bench shapes (profiles/traffic.json is keyed by the kernel sources' hash), the long-read
# table with kernel stats and PMC of the cooperative kernel, the per-read hosts
cd ${GRAFT_REPO_ROOT:-/root/repo}
step=${1:-all}
if [ $step = all ] || [ $step = prof ]; then bash tools/r6_final.sh prof; fi
if [ $step = all ] || [ $step = long ]; then bash tools/r6_final.sh long; fi
if [ $step = all ] || [ $step = perread ]; then bash tools/r6_per_read2.sh > gpurun_out/r6_per_read3.txt 2>&1; tail -5 gpurun_out/r6_per_read3.txt; fi
