#!/bin/bash
# seed-hit stage: parity tests, the bench leg (worst case: every read full of equal x) and the kernels of the end-to-end batched host
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
export MM2C_QUIET=1
timeout -k 10 600 python -m pytest tests/test_gpu_seed_hits.py -m gpu -x -q 2>&1 | tail -5 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tie or equal_first or chains" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --no-e2e --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seed_hits', d['seed_hits']['value']/1e9, d['seed_hits']['ms'], 'epilogue_ms', d['whole_mm_chain_dp']['epilogue_ms'])" || exit 1
bash tools/r4_e2e_kernels.sh 50 120000 2>&1 | grep -E "seed_|tiesort" | cut -d, -f1-4
