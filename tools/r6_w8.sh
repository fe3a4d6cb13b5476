cd /root/repo
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python3 -m pytest tests/test_gpu_long_reads.py tests/test_gpu_parity.py tests/test_gpu_multidevice.py -x -q -m gpu > gpurun_out/fuse_tests.txt 2>&1 || { tail -40 gpurun_out/fuse_tests.txt; exit 1; }
tail -3 gpurun_out/fuse_tests.txt
MM2C_SOAK_SECONDS=60 timeout -k 10 300 python3 tools/soak.py 100000 950000 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/r6_per_read2.sh > gpurun_out/r6_per_read3.txt 2>&1; grep "wall\|per call\|staged passes:" gpurun_out/r6_per_read3.txt | cut -c1-250
