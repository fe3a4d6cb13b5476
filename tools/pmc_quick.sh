#!/bin/bash
# Runs on the GPU box: two PMC passes (instruction counts, issue activity) of bench.py for quick kernel iteration.
# usage: tools/pmc_quick.sh <tag> [bench args...]      output: gpurun_out/pmcq/<tag>/summary.txt
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcq/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary $*"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc $i exit $?" >> $OUT/log.txt
done
python3 - "$OUT" "$TAG" <<'PY' > $OUT/summary.txt
import csv, glob, sys, collections, os
src, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for fn in glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "chain_dp" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(tag, k)
    for c, v in sorted(d.items()):
        v = sorted(v)[len(v)//2:]           # main pass (the flagged-task redo pass of the same name counts ~0)
        print(f"  {c:28s} {sum(v)/len(v):14.6g}")
PY
cat $OUT/summary.txt
