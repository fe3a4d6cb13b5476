#!/bin/bash
# round 6, final evidence: the whole GPU test suite, two soaks, the driver's line
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/r6_final; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gpu_suite.txt; cat $O/gpu_suite.txt
MM2C_SOAK_SECONDS=150 timeout -k 10 400 python3 tools/soak.py 100000 600000 2>&1 | grep -v amdgpu.ids | tail -4 > $O/soak.txt; cat $O/soak.txt
MM2C_SOAK_SECONDS=100 MM2C_SOAK_CUT=64 timeout -k 10 300 python3 tools/soak.py 100000 700000 2>&1 | grep -v amdgpu.ids | tail -4 > $O/soak_cut.txt; cat $O/soak_cut.txt
MM2C_SOAK_SECONDS=80 timeout -k 10 300 python3 tools/soak2.py 100000 800000 2>&1 | grep -v amdgpu.ids | tail -4 > $O/soak2.txt; cat $O/soak2.txt
timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
