#!/bin/bash
# round 5: timeline of the per-read passes (kernel + copy trace of a short path-B run) and the fixed costs of the process (init, shutdown)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/pathb; mkdir -p $W $REPO/gpurun_out/r5_trace
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb 50 --reads 120000 > /dev/null 2>&1 || exit 1
head -n 16000 $W/syn.reads.fa > $W/few.reads.fa
export MM2_MINI_BATCH=100000000 MM2_TIMING=1
for RUN in 1 2; do
  T0=$(date +%s.%N); MM2C_PASS_TIMING=1 timeout -k 10 300 $REPO/oracle/_ref/mm2_gpuhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/b.paf 2> $W/b.err; T1=$(date +%s.%N)
  echo "path B run $RUN: wall $(python3 -c "print(round($T1-$T0,2))") s"; grep -E "mm2c_init|mm2c_shutdown|staged passes|requests per" $W/b.err | cut -c1-300
done > $REPO/gpurun_out/r5_trace/fixed_costs.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $W/prof -o pb -- $REPO/oracle/_ref/mm2_gpuhost -t 16 $W/syn.ref.fa $W/few.reads.fa > $W/c.paf 2> $W/c.err
echo "profiled run rc $?" >> $REPO/gpurun_out/r5_trace/fixed_costs.txt
for f in $(find $W/prof -name "*.csv"); do cp $f $REPO/gpurun_out/r5_trace/; done
ls -la $REPO/gpurun_out/r5_trace/
