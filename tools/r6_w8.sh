cd /root/repo
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python3 -m pytest tests/test_gpu_long_reads.py -x -q -m gpu > gpurun_out/fuse_tests.txt 2>&1 || { tail -40 gpurun_out/fuse_tests.txt; exit 1; }
tail -3 gpurun_out/fuse_tests.txt
bash tools/r6_per_read2.sh > gpurun_out/r6_per_read3.txt 2>&1; grep "wall\|per call\|staged passes:" gpurun_out/r6_per_read3.txt | cut -c1-250
