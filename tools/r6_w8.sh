cd /root/repo
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python3 -m pytest tests/test_gpu_long_reads.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/nxt_tests.txt 2>&1 || { tail -30 gpurun_out/nxt_tests.txt; exit 1; }
tail -3 gpurun_out/nxt_tests.txt
timeout -k 10 600 python tools/long_reads.py --no-seed --reps 3 --routes auto --sizes 2048x100000,1024x300000,256x1000000,600x20000,200x20000 2>&1 | grep "^==\|^DP" > gpurun_out/nxt_long.txt; cat gpurun_out/nxt_long.txt
