#!/bin/bash
# where the time of seed_ties goes on the worst-case batch (every read full of equal x): kernel cut after phases (MM2C_TIE_CUT; results are then wrong, the probe only times)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
export MM2C_QUIET=1
for CUT in 0 1 2 3; do echo "cut $CUT: $(MM2C_TIE_CUT=$CUT timeout -k 10 200 python tools/seed_probe.py 8192 5000 mixed 2>/dev/null | grep 'seed hits ->')"; done
