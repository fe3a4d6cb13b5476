cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "several_waves or hand_written_loop_equals or fpga_v2 or host_paths or reference_symbol or concurrent_callers or one_long_task" 2>&1 | tail -5 || exit 1
timeout -k 10 400 python3 tools/long_reads.py --no-seed --routes coop16 2>&1 | grep -v "^#\|amdgpu.ids"
MM2C_LIB_PATH=$PWD/minimap2-fpga_amd/variants/coop_n9.so timeout -k 10 120 python3 tools/long_reads.py --no-seed --routes coop16 --sizes 256x200000 --reps 1 --check 1 2>&1 | grep "ticks"
