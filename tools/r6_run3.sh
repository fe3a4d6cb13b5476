cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for v in coop_n9; do
  echo "=== $v"; MM2C_LIB_PATH=$PWD/minimap2-fpga_amd/variants/$v.so timeout -k 10 120 python3 tools/long_reads.py --no-seed --routes coop16 --sizes 256x200000 --reps 2 --check 1 2>&1 | grep -v "^#\|amdgpu.ids" || true
done
LONG_PROFILE=0 LONG_PMC="--routes coop16 --sizes 256x200000" timeout -k 10 300 tools/long_reads.sh r6_rows3 --routes coop16 --sizes 256x200000 --no-seed 2>&1 | grep -v "amdgpu.ids" | grep -A18 "^mm2c::chain_dp_coop"
