#!/bin/bash
# per-kernel times of the batched host (INTEGRATION.md path C) on the end-to-end workload: rocprofv3 --kernel-trace --stats around mm2_batchhost
# usage: tools/r4_e2e_kernels.sh [genome_mb] [reads]   -> gpurun_out/e2e_kernels/kernel_stats.csv
GMB=${1:-50}; READS=${2:-120000}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/e2e_kernels; mkdir -p $OUT
W=/tmp/e2e_kernels; mkdir -p $W
export TMPDIR=/tmp MM2_MINI_BATCH=${MM2_MINI_BATCH:-100000000}
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb $GMB --reads $READS > $OUT/gen.log 2>&1 || exit 1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $W/prof -o run --output-format csv -- $REPO/oracle/_ref/mm2_batchhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/batch.paf 2> $OUT/batch.err || exit 1
cp $(find $W/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
md5sum $W/batch.paf | cut -c1-32 > $OUT/paf.md5
head -30 $OUT/kernel_stats.csv | cut -c1-200
