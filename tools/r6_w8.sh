cd /root/repo
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python3 -m pytest tests/test_gpu_seed_hits.py -x -q -m gpu > gpurun_out/tie_tests.txt 2>&1 || { tail -40 gpurun_out/tie_tests.txt; exit 1; }
tail -2 gpurun_out/tie_tests.txt
timeout -k 10 600 python tools/long_reads.py --no-dp --reps 3 2>&1 | grep "^==\|^seed" | cut -c1-200
