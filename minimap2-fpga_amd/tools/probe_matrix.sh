#!/bin/bash
# On the GPU box: DP kernel ms of several streams for several library variants on ONE box.   usage: tools/probe_matrix.sh name [name ...]   (base = the in-tree library;
# base0 = the in-tree library with MM2C_COMPACT_RING=0; base_s0 = with MM2C_SPLIT_STREAMS=0); MATRIX=ragged: the ragged streams instead of the usual five
if [ "${MATRIX:-}" = ragged ]; then
  STREAMS=("--profile mixed --ragged" "--profile dense --ragged" "--preset asm20 --profile mixed --ragged" "--preset ava-ont --profile mixed" "--profile mixed")
else
  STREAMS=("--profile mixed" "--profile dense" "--preset asm20 --profile mixed" "--profile mixed --ragged" "--profile colinear")
fi
for ARGS in "${STREAMS[@]}"; do
  for NAME in "$@"; do
    if [ "$NAME" = base_s0 ]; then MM2C_SPLIT_STREAMS=0 tools/probe_run.sh "$ARGS" base | sed "s/^base/base_s0 [$ARGS]/"
    elif [ "$NAME" = base0 ]; then MM2C_COMPACT_RING=0 tools/probe_run.sh "$ARGS" base | sed "s/^base/base0 [$ARGS]/"
    else tools/probe_run.sh "$ARGS" $NAME | sed "s/^$NAME/$NAME [$ARGS]/"; fi
  done
done
