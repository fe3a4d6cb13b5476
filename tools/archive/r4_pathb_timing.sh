#!/bin/bash
# path B (one synchronous call per read) on the end-to-end workload with the pass timers of the call combiner on (MM2C_PASS_TIMING=1)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/pathb; mkdir -p $W $REPO/gpurun_out
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb 50 --reads 120000 > /dev/null 2>&1 || exit 1
export MM2_MINI_BATCH=100000000
for RUN in 1 2; do
  T0=$(date +%s.%N)
  MM2C_PASS_TIMING=1 timeout -k 10 300 $REPO/oracle/_ref/mm2_gpuhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/b.paf 2> $W/b.err
  T1=$(date +%s.%N)
  echo "run $RUN: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/b.paf | cut -c1-8)"; grep -E "staged passes|per call|passes" $W/b.err | cut -c1-300
done
