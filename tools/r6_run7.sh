cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for cut in 1 2 3 0; do echo "== MM2C_TIE_CUT=$cut"; MM2C_TIE_CUT=$cut timeout -k 10 300 python3 tools/long_reads.py --no-dp --sizes 2048x100000,256x1000000 --reps 2 --check 1 2>&1 | grep "seed hits"; done
