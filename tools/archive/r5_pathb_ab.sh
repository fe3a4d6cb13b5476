#!/bin/bash
# round 5: path B (one synchronous call per read) on the end-to-end workload, A/B over the knobs named on the command line ("VAR=val VAR=val" per run)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/pathb; mkdir -p $W $REPO/gpurun_out
[ -f $W/syn.reads.fa ] || python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb 50 --reads 120000 > /dev/null 2>&1 || exit 1
export MM2_MINI_BATCH=100000000 MM2_TIMING=1
T0=$(date +%s.%N); timeout -k 10 300 $REPO/oracle/_ref/mm2_refhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/a.paf 2> $W/a.err; T1=$(date +%s.%N)
echo "cpu host: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/a.paf | cut -c1-8)"
for CFG in "$@"; do
  for RUN in 1 2; do
    T0=$(date +%s.%N)
    env $CFG MM2C_PASS_TIMING=1 timeout -k 10 300 $REPO/oracle/_ref/mm2_gpuhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/b.paf 2> $W/b.err
    T1=$(date +%s.%N)
    echo "path B [$CFG] run $RUN: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/b.paf | cut -c1-8)"; grep -E "staged passes|requests per|per call|per device slot|ERROR|rror" $W/b.err | cut -c1-400
  done
done
for RUN in 1 2; do
  T0=$(date +%s.%N); timeout -k 10 300 $REPO/oracle/_ref/mm2_splithost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/s.paf 2> $W/s.err; T1=$(date +%s.%N)
  echo "split host (path A restated) run $RUN: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/s.paf | cut -c1-8)"; grep -E "split model|per call" $W/s.err | cut -c1-400
done
T0=$(date +%s.%N); timeout -k 10 300 $REPO/oracle/_ref/mm2_refhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/a.paf 2> $W/a.err; T1=$(date +%s.%N)
echo "cpu host again: wall $(python3 -c "print(round($T1-$T0,2))") s"
