#!/bin/bash
# round 6, second pass over the per-read hosts: path B as one launch per pass (the default), with stage_in in front (MM2C_SINGLE_LAUNCH=0) and with the prepass launch as well (MM2C_FUSE_ST=0)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/pathb; mkdir -p $W $REPO/gpurun_out
export GPU_MAX_HW_QUEUES=16
[ -f $W/syn.reads.fa ] || python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb 50 --reads 120000 > /dev/null 2>&1 || exit 1
export MM2_MINI_BATCH=100000000 MM2_TIMING=1 MM2C_QUIET=1
run() {  # name exe env...
  local name=$1 exe=$2; shift 2
  for RUN in 1 2; do
    T0=$(date +%s.%N); env "$@" MM2C_PASS_TIMING=1 timeout -k 10 300 $REPO/oracle/_ref/$exe -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/x.paf 2> $W/x.err; T1=$(date +%s.%N)
    echo "$name run $RUN: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/x.paf | cut -c1-8)"
    grep -E "staged passes|requests per|GPU chaining|split model|ERROR|rror" $W/x.err | cut -c1-330
  done
}
run "cpu host" mm2_refhost A=1
run "path B, one launch per pass (default)" mm2_gpuhost A=1
run "path B, stage_in + kernel with its own window starts" mm2_gpuhost MM2C_SINGLE_LAUNCH=0
run "path B, stage_in + prepass + kernel (round 6's first form)" mm2_gpuhost MM2C_SINGLE_LAUNCH=0 MM2C_FUSE_ST=0
run "split host, never decline (default)" mm2_splithost MM2C_DECLINE_WHEN_BUSY=0
run "cpu host again" mm2_refhost A=1
