cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python3 -m pytest tests/test_gpu_seed_hits.py tests/test_gpu_parity.py -x -q -k "seed or epilogue or replays or soak" 2>&1 | tail -4 || exit 1
echo "== default"; timeout -k 10 300 python3 tools/long_reads.py --no-dp 2>&1 | grep -v "^#\|amdgpu.ids"
