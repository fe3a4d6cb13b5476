#!/bin/bash
# kernel stats of the secondary legs of bench.py (device epilogue, seed hits, host-streamed), one rocprofv3 --kernel-trace --stats run
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof/r4_secondary; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-e2e > $OUT/run.log 2>&1
cd $REPO
python3 - <<PY
import csv,glob
for fn in glob.glob("gpurun_out/prof/r4_secondary/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(fn)):
        if "mm2c::" in r["Name"] and float(r["TotalDurationNs"])>2e5:
            print("%-72s calls %4s avg %8.3f ms total %8.2f ms" % (r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")[:72], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY
