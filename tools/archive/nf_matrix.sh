#!/bin/bash
# DP kernel time of the streams that use the long-ring class, for library variants built by tools/probe_build.sh (GPU box): tools/nf_matrix.sh name [name ...]
for ARGS in "--profile dense" "--preset ava-ont" "--profile dense --preset asm20" "--preset ava-ont --profile colinear"; do echo "== $ARGS"; tools/probe_run.sh "$ARGS" "$@"; done
