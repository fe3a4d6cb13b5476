#!/bin/bash
# per-kernel times of the device epilogue on the headline batch (rocprofv3 kernel stats): tools/epi_profile.sh <tag> [profile]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-epi}; PROF=${2:-mixed}
export TMPDIR=/tmp
cd /tmp
OUT=$REPO/gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/epilogue_probe.py 65536 5000 $PROF --device-only > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
grep -E "epi_|chain_dp_tile<8, 2, true, false|chain_window" $f | awk -F, '{printf "%-70s calls %s avg_ms %.3f\n", substr($1,1,70), $2, $4/1e6}'
grep -i "epilogue\|anchors/s" $OUT/log.txt | tail -6
