cd /root/repo
export GPU_MAX_HW_QUEUES=16
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/w8_suite.txt 2>&1; tail -8 gpurun_out/w8_suite.txt
MM2C_SOAK_SECONDS=100 timeout -k 10 300 python3 tools/soak.py 100000 600000 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/w8_soak.txt; cat gpurun_out/w8_soak.txt
