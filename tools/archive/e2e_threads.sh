#!/bin/bash
# path B (one synchronous library call per read) with more host threads than cores: every thread beyond the cores only adds chaining calls in flight.
# usage: tools/e2e_threads.sh [genome_mb] [reads] "<thread counts>"
GMB=${1:-50}; READS=${2:-120000}; TL=${3:-"16 48 96 192"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/e2e_threads; mkdir -p $OUT; rm -f $OUT/summary.txt
W=/tmp/e2e_threads; mkdir -p $W
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb $GMB --reads $READS > $OUT/gen.log 2>&1
run() {
  local name=$1 exe=$2 T=$3
  local T0=$(date +%s.%N)
  timeout -k 10 300 $REPO/oracle/_ref/$exe -t $T $W/syn.ref.fa $W/syn.reads.fa > $W/$name.paf 2> $OUT/$name.err
  local RC=$?
  local T1=$(date +%s.%N)
  echo "$name -t $T exit $RC wall $(python3 -c "print(round($T1-$T0,2))") s lines $(wc -l < $W/$name.paf) md5 $(md5sum < $W/$name.paf | cut -c1-32)" >> $OUT/summary.txt
}
for T in $TL; do run ref_t$T mm2_refhost $T; run gpuhost_t$T mm2_gpuhost $T; done
run batch mm2_batchhost 16
cat $OUT/summary.txt
