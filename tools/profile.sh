#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then PMC passes (each in its own run,
# never combined with tracing domains other than --kernel-trace).  Output: gpurun_out/prof/<tag>/
set -u
TAG=${1:-r1}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace exit $?" >> $OUT/trace.log
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1
  echo "pmc $name exit $?" >> $OUT/trace.log
done
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
