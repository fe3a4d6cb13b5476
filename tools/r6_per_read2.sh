#!/bin/bash
# round 6, second pass over the per-read hosts: path B with the window starts made inside the cooperative kernel (MM2C_FUSE_ST=1, the default) and by the prepass launch (0)
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
run "path B, window starts in the kernel" mm2_gpuhost MM2C_FUSE_ST=1
run "path B, prepass launch" mm2_gpuhost MM2C_FUSE_ST=0
run "split host, never decline, window starts in the kernel" mm2_splithost MM2C_DECLINE_WHEN_BUSY=0
run "cpu host again" mm2_refhost A=1
