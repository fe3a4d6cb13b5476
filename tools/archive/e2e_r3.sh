#!/bin/bash
# Round-3 end-to-end comparison on the GPU box (config-3 stand-in): the three hosts of tools/e2e_batch.sh at two mini-batch sizes (-K), the
# batched host with the index's position arrays resident on the GPU (default) and with the hits copied per batch (MM2_BATCH_HOSTPOOL=1), and with page-locked instead of plain chain outputs (MM2_BATCH_PINNED_OUT=1).
# usage: tools/e2e_r3.sh [genome_mb] [reads] [threads] [tag]
GMB=${1:-50}; READS=${2:-120000}; THREADS=${3:-16}; TAG=${4:-r3}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/e2e_$TAG; mkdir -p $OUT; rm -f $OUT/summary.txt
W=/tmp/e2e_$TAG; mkdir -p $W
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb $GMB --reads $READS > $OUT/gen.log 2>&1
echo "genome ${GMB} Mb, ${READS} reads, ${THREADS} threads" >> $OUT/summary.txt
run() {   # name exe K hostpool [pageable_out]
  local name=$1 exe=$2 K=$3 HP=$4
  export MM2_BATCH_PINNED_OUT=${5:-0}
  local T0=$(date +%s.%N)
  MM2_MINI_BATCH=$K MM2_BATCH_HOSTPOOL=$HP timeout -k 10 300 $REPO/oracle/_ref/$exe -t $THREADS $W/syn.ref.fa $W/syn.reads.fa > $W/$name.paf 2> $OUT/$name.err
  local RC=$?
  local T1=$(date +%s.%N)
  echo "$name exit $RC wall $(python3 -c "print(round($T1-$T0,2))") s lines $(wc -l < $W/$name.paf) md5 $(md5sum < $W/$name.paf | cut -c1-32)" >> $OUT/summary.txt
  grep "mm2_batchhost\]" $OUT/$name.err >> $OUT/summary.txt
}
for K in 500000000 100000000; do
  run ref_K$K mm2_refhost $K 0
  run batch_pool_K$K mm2_batchhost $K 0
  run batch_hostpool_K$K mm2_batchhost $K 1
  run batch_pool_pinnedout_K$K mm2_batchhost $K 0 1
done
run gpuhost_K500000000 mm2_gpuhost 500000000 0
run batch_pool_K500000000_again mm2_batchhost 500000000 0
cat $OUT/summary.txt
