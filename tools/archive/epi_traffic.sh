#!/bin/bash
# HBM traffic of the device epilogue on the headline batch: FETCH_SIZE and WRITE_SIZE in separate PMC passes (never combined with other
# trace domains), summed over the epilogue kernels of one pass.   tools/epi_traffic.sh <tag> [profile]   (MM2C_EPI_FUSED=0: kernels A / B / C)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-epitraffic}; PROF=${2:-mixed}
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$REPO/gpurun_out/prof/${TAG}_$c
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/epilogue_probe.py 65536 5000 $PROF --device-only > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
done
python3 - "$REPO/gpurun_out/prof" "$TAG" <<'PY'
import csv, glob, sys, collections, os
root, tag = sys.argv[1], sys.argv[2]
A = 65536 * 5000
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(os.path.join(root, f"{tag}_{c}", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "epi_" not in k: continue
            tot[k.split("(anonymous namespace)::")[-1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
s = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for k, d in sorted(tot.items()):
    row = []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = d.get(c, [0]); m = sum(v) / len(v) * 1024 / A; s[c] += m; row.append(f"{c} {m:7.2f} B/anchor")
    print(f"{k:24s} " + "  ".join(row))
print(f"{'epilogue total':24s} FETCH {s['FETCH_SIZE']:.1f}  WRITE {s['WRITE_SIZE']:.1f}  sum {s['FETCH_SIZE'] + s['WRITE_SIZE']:.1f} B/anchor (FETCH_SIZE as counted: 16-byte loads are tallied at half their bytes on gfx950)")
PY
