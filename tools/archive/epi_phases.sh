#!/bin/bash
# development aid: time of the fused epilogue kernel cut after phase N (MM2C_EPI_PHASES: 1 load + marks, 2 v / peaks, 3 chain ends, 4 sort,
# 5 owners, 6 depths, 7 chains, 0 everything)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for ph in 1 2 3 4 5 6 7 0; do
  OUT=$REPO/gpurun_out/prof/epiph_$ph
  rm -rf $OUT; mkdir -p $OUT
  MM2C_EPI_PHASES=$ph timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/tools/epilogue_probe.py 65536 5000 ${1:-mixed} --device-only > $OUT/log.txt 2>&1 || exit 1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== phases=$ph"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "epi_" in r["Name"]:
        print("   %-50s calls %s avg_ms %.3f" % (r["Name"].split("(anonymous namespace)::")[-1][:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
