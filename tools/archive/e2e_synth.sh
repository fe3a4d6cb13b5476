#!/bin/bash
# End-to-end map-ont on a synthetic genome (BASELINE config 3 stand-in), on the GPU box:
#   reference host objects + CPU oracle chaining (mm2_refhost)  vs  reference host objects + GPU chaining (mm2_gpuhost)
# usage: tools/e2e_synth.sh [genome_mb] [reads] [threads]
GMB=${1:-50}; READS=${2:-5000}; THREADS=${3:-16}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/e2e; mkdir -p $OUT
W=/tmp/e2e_synth; mkdir -p $W
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb $GMB --reads $READS > $OUT/gen.log 2>&1
for exe in mm2_refhost mm2_gpuhost; do
  T0=$(date +%s.%N)
  $REPO/oracle/_ref/$exe -t $THREADS $W/syn.ref.fa $W/syn.reads.fa > $W/$exe.paf 2> $OUT/$exe.err
  RC=$?
  T1=$(date +%s.%N)
  echo "$exe exit $RC wall $(python3 -c "print(round($T1-$T0,2))") s lines $(wc -l < $W/$exe.paf) md5 $(md5sum < $W/$exe.paf | cut -c1-32)" >> $OUT/summary.txt
done
cmp $W/mm2_refhost.paf $W/mm2_gpuhost.paf && echo "PAF identical (genome ${GMB} Mb, ${READS} reads, ${THREADS} threads)" >> $OUT/summary.txt
tail -3 $OUT/mm2_gpuhost.err >> $OUT/summary.txt
cat $OUT/summary.txt
