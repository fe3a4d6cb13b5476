#!/bin/bash
# End-to-end map-ont on a synthetic genome, on the GPU box, three hosts built from the reference's own objects:
#   mm2_refhost   = CPU chaining (oracle mm_chain_dp), one read at a time per thread
#   mm2_gpuhost   = the library's mm_chain_dp, one synchronous GPU call per read (drop-in path)
#   mm2_batchhost = worker_for restructured: seed all -> ONE GPU call (matches in, chains out) -> post all  
# usage: tools/e2e_batch.sh [genome_mb] [reads] [threads] [mini_batch_bases]   (mini-batch = minimap2's -K, default 500M, same for all hosts)
GMB=${1:-50}; READS=${2:-5000}; THREADS=${3:-16}
if [ -n "${4:-}" ]; then export MM2_MINI_BATCH=$4; fi
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/e2e_batch; mkdir -p $OUT
W=/tmp/e2e_batch; mkdir -p $W
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb $GMB --reads $READS > $OUT/gen.log 2>&1
for exe in mm2_refhost mm2_gpuhost mm2_batchhost; do
  T0=$(date +%s.%N)
  timeout -k 10 600 $REPO/oracle/_ref/$exe -t $THREADS $W/syn.ref.fa $W/syn.reads.fa > $W/$exe.paf 2> $OUT/$exe.err
  RC=$?
  T1=$(date +%s.%N)
  echo "$exe exit $RC wall $(python3 -c "print(round($T1-$T0,2))") s lines $(wc -l < $W/$exe.paf) md5 $(md5sum < $W/$exe.paf | cut -c1-32)" >> $OUT/summary.txt
done
cmp $W/mm2_refhost.paf $W/mm2_batchhost.paf && echo "PAF of the batched host identical to the CPU host's (genome ${GMB} Mb, ${READS} reads, ${THREADS} threads)" >> $OUT/summary.txt
tail -1 $OUT/mm2_batchhost.err >> $OUT/summary.txt
cat $OUT/summary.txt
