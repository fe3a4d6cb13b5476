#!/bin/bash
# round 6: the per-read hosts on the end-to-end workload (50 Mb synthetic genome, 120 000 reads, -t 16): path B (every read to the GPU), and the split host (path A restated:
# chain.c's HW / SW decision + the busy protocol) under the three decline rules of mm2c_chain_task_host_pred (MM2C_DECLINE_WHEN_BUSY 0 never / 1 measured / 2 round 5's).
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
run "path B" mm2_gpuhost A=1
run "split host, never decline (default)" mm2_splithost MM2C_DECLINE_WHEN_BUSY=0
run "split host, decline by measured service time" mm2_splithost MM2C_DECLINE_WHEN_BUSY=1
run "split host, round 5 rule (booked predictions)" mm2_splithost MM2C_DECLINE_WHEN_BUSY=2
run "split host, everything to the device" mm2_splithost MM2_SPLIT_ALL_HW=1
run "cpu host again" mm2_refhost A=1
