#!/bin/bash
# round 5 (verdict item 8): the config-3 stand-in once at SURVEY 8d's scale -- a 1 Gb synthetic genome, 100 000 simulated ONT reads -- through the three hosts
# (CPU chaining, path C batched, path B per read), same -t / -K; PAF md5, stage sums, hit-pool size and upload time.  Recorded in profiles/r5_e2e_1g.md.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
GMB=${1:-1000}; READS=${2:-100000}; T=${3:-16}
W=/tmp/e2e1g; mkdir -p $W $REPO/gpurun_out
OUT=$REPO/gpurun_out/r5_e2e_1g.txt
{
echo "genome ${GMB} Mb, ${READS} reads, -t $T, -K 100M; $(nproc) cores visible, $(free -g | awk '/Mem:/{print $2}') GiB host memory"
T0=$(date +%s.%N); python3 $REPO/tools/make_synth_genome.py $W/syn --genome-mb $GMB --reads $READS > /dev/null 2>&1 || exit 1; T1=$(date +%s.%N)
echo "generated in $(python3 -c "print(round($T1-$T0,1))") s: $(du -h $W/syn.ref.fa | cut -f1) reference, $(du -h $W/syn.reads.fa | cut -f1) reads"
export MM2_MINI_BATCH=100000000 MM2_TIMING=1
for H in mm2_refhost mm2_batchhost mm2_gpuhost; do
  T0=$(date +%s.%N)
  timeout -k 10 900 $REPO/oracle/_ref/$H -t $T $W/syn.ref.fa $W/syn.reads.fa > $W/$H.paf 2> $W/$H.err
  RC=$?; T1=$(date +%s.%N)
  echo "== $H: rc $RC wall $(python3 -c "print(round($T1-$T0,2))") s, PAF $(wc -l < $W/$H.paf) lines md5 $(md5sum < $W/$H.paf | cut -c1-32)"
  grep -E "stages \(summed|inside the library|batched GPU calls|GPU chaining|per device slot|position arrays|mm2c_init|HIP start-up|M::mm_idx_gen|M::worker_pipeline::" $W/$H.err | cut -c1-420 | tail -12
done
} > $OUT 2>&1
cat $OUT
