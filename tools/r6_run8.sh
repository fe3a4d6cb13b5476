cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python3 -m pytest tests/test_gpu_multidevice.py tests/test_gpu_dropin_e2e.py -x -q 2>&1 | tail -4 || exit 1
timeout -k 10 700 tools/r6_per_read.sh 2>&1 | tee gpurun_out/r6_per_read.txt | tail -60
