set -x
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for v in coop_p9 coop_p1 coop_p2 coop_p3 coop_p4; do
  echo "=== $v"; MM2C_LIB_PATH=$PWD/minimap2-fpga_amd/variants/$v.so timeout -k 10 120 python3 tools/long_reads.py --no-seed --routes coop16 --sizes 256x200000 --reps 2 --check 1 2>&1 | grep -v "^#" || true
done
LONG_PROFILE=1 timeout -k 10 400 tools/long_reads.sh r6_seedbase --no-dp --sizes 2048x100000,256x1000000 2>&1 | tail -40
timeout -k 10 600 python3 -m pytest tests/test_gpu_multidevice.py -x -q 2>&1 | tail -15
