#!/bin/bash
# development aid: per-kernel times of the device epilogue with the kernels cut after phase N (MM2C_EPI_PHASES)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for ph in 0 1 2 3 11; do
  OUT=$REPO/gpurun_out/prof/epiph_$ph
  rm -rf $OUT; mkdir -p $OUT
  MM2C_EPI_PHASES=$ph timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/epilogue_probe.py 65536 5000 ${1:-mixed} --device-only > $OUT/log.txt 2>&1 || exit 1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== phases=$ph"; grep -E "epi_|chain_dp_wave<256, true, false" $f | awk -F, '{printf "%-60s avg_ms %.3f\n", substr($1,1,60), $4/1e6}'
done
