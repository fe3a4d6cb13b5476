#!/bin/bash
# round 5: where a per-read call (INTEGRATION.md path B) spends its time on the end-to-end workload, beside what the same calls cost on the CPU
# (oracle/ref_host/chain_shim.c, MM2O_TIME=1); then the kernel / copy durations of the same run by rocprofv3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/pathb; mkdir -p $W $REPO/gpurun_out
OUT=$REPO/gpurun_out/r5_pathb_probe.txt
python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb 50 --reads 120000 > /dev/null 2>&1 || exit 1
export MM2_MINI_BATCH=100000000
{
T0=$(date +%s.%N); MM2O_TIME=1 timeout -k 10 300 $REPO/oracle/_ref/mm2_refhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/a.paf 2> $W/a.err; T1=$(date +%s.%N)
echo "cpu host: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/a.paf | cut -c1-8)"; grep chain_shim $W/a.err
for RUN in 1 2; do
  T0=$(date +%s.%N)
  MM2C_PASS_TIMING=1 timeout -k 10 300 $REPO/oracle/_ref/mm2_gpuhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/b.paf 2> $W/b.err
  T1=$(date +%s.%N)
  echo "path B run $RUN: wall $(python3 -c "print(round($T1-$T0,2))") s md5 $(md5sum < $W/b.paf | cut -c1-8)"; grep -E "mm2chain|per call|passes" $W/b.err | cut -c1-300
done
} > $OUT 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $W/prof -o pb -- $REPO/oracle/_ref/mm2_gpuhost -t 16 $W/syn.ref.fa $W/syn.reads.fa > $W/c.paf 2> $W/c.err
echo "profiled run rc $?" >> $OUT
for f in $(find $W/prof -name "*stats*.csv" | sort); do echo "== $f" >> $OUT; head -25 $f >> $OUT; done
tail -5 $W/c.err >> $OUT
