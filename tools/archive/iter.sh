#!/bin/bash
# one kernel-iteration round on the GPU box: parity tests, quick bench of three profiles, instruction counters of the headline stream
# usage: tools/iter.sh <tag> [pmc=1]
TAG=$1; PMC=${2:-1}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/iter_$TAG.tests.log 2>&1
tail -2 gpurun_out/iter_$TAG.tests.log
rm -f gpurun_out/quick.log
tools/quickbench.sh $TAG "--no-secondary" mixed dense colinear
tools/quickbench.sh $TAG-ava "--no-secondary --preset ava-ont" colinear
cat gpurun_out/quick.log
if [ "$PMC" = "1" ]; then
  tools/pmc_quick.sh $TAG > /dev/null 2>&1
  python3 - <<PY
import re
d={}
for l in open("gpurun_out/pmcq/$TAG/summary.txt"):
    if l.startswith("$TAG"):
        if d: break
        continue
    k,v=l.split(); d[k]=float(v)
A=3.2768e8
print("per anchor: " + "  ".join(f"{k[3:].replace('INSTS_','')} {d[k]/A:.1f}" for k in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_BRANCH","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_FLAT","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_ANY") if k in d))
PY
fi
