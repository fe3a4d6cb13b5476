cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_zz_timing.py 2>&1 | tail -40
