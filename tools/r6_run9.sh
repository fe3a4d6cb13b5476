cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout -k 10 300 python3 tools/r6_hoststream.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_hoststream.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_multidevice.py tests/test_gpu_parity.py -x -q -k "busy or pipelined or host" 2>&1 | tail -3
