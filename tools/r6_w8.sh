cd /root/repo
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python3 -m pytest tests/test_gpu_long_reads.py tests/test_gpu_parity.py tests/test_gpu_multidevice.py -x -q -m gpu > gpurun_out/fuse_tests.txt 2>&1 || { tail -40 gpurun_out/fuse_tests.txt; exit 1; }
tail -3 gpurun_out/fuse_tests.txt
timeout -k 10 600 python tools/long_reads.py --no-seed --reps 2 --routes auto --sizes 256x1000000,1024x300000 2>&1 | grep "^DP"
bash tools/r6_per_read2.sh > gpurun_out/r6_per_read3.txt 2>&1; grep "wall\|per call\|staged passes:" gpurun_out/r6_per_read3.txt | cut -c1-250
