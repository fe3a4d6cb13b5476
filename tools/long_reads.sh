#!/bin/bash
# Runs on the GPU box (via gpurun): the long-read streams of tools/long_reads.py (DP routes + seed hits), then rocprofv3 kernel stats of the same command
# and two PMC passes of the DP alone.  usage: tools/long_reads.sh <tag> [long_reads.py args]     output: gpurun_out/long/<tag>/
set -u
TAG=${1:-r6}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/long/$TAG
mkdir -p $OUT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
cd /tmp
python3 $REPO/tools/long_reads.py "$@" > $OUT/run.txt 2>&1 || { echo "long_reads.py failed" >> $OUT/run.txt; cat $OUT/run.txt; exit 1; }
cat $OUT/run.txt
if [ "${LONG_PROFILE:-1}" = "1" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/long_reads.py --reps 1 "$@" > $OUT/trace.log 2>&1
  echo "trace exit $?" >> $OUT/trace.log
  for f in $OUT/trace/*/*_kernel_stats.csv; do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mm2c::" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    name = r['Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    print(f"{name[:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:10.3f} ms  total {float(r['TotalDurationNs'])/1e6:10.2f} ms")
PY
  done > $OUT/kernel_stats.txt
  cat $OUT/kernel_stats.txt
fi
if [ -n "${LONG_PMC:-}" ]; then
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $REPO/tools/long_reads.py --reps 1 --no-seed $LONG_PMC > $OUT/pmc$i.log 2>&1
    echo "pmc $i exit $?" >> $OUT/trace.log
  done
  python3 - "$OUT" <<'PY' > $OUT/pmc_summary.txt
import csv, glob, sys, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(sys.argv[1], "pmc*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "chain_dp" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} max per launch {max(v):14.6g}  (launches {len(v)})")
PY
  cat $OUT/pmc_summary.txt
fi
