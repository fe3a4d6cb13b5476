#!/bin/bash
# PMC counters of the seed-hit kernels on the driver line's worst-case batch (tools/seed_probe.py 8192 5000 mixed): two rocprofv3 --pmc passes (instruction counts, issue
# activity), per kernel and per anchor.   output: gpurun_out/seed_pmc/summary.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/seed_pmc; mkdir -p $OUT
export TMPDIR=/tmp MM2C_QUIET=1
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $REPO/tools/seed_probe.py 8192 5000 mixed > $OUT/pmc$i.log 2>&1
  echo "pmc $i exit $?" >> $OUT/log.txt
done
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import csv, glob, sys, collections, os
src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "seed_" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
A = 40960000.0
for k, d in sorted(agg.items()):
    print("### `%s` (mean per launch; per anchor of the 4.096e7 of the batch)" % k)
    print("| counter | per launch | per anchor |\n|---|---|---|")
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print("| %s | %.4g | %.4g |" % (c, m, m / A))
    print()
PY
cat $OUT/summary.txt | head -60
