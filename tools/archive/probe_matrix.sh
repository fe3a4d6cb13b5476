#!/bin/bash
# On the GPU box: DP kernel ms of several streams for several library variants on ONE box.   usage: tools/probe_matrix.sh name [name ...]   (base = the in-tree library;
# base0 = the in-tree library with MM2C_COMPACT_RING=0)
for ARGS in ${MATRIX_ARGS:-"--profile mixed" "--profile dense" "--preset asm20 --profile mixed" "--profile mixed --ragged" "--profile colinear"}; do
  for NAME in "$@"; do
    if [ "$NAME" = base_s0 ]; then MM2C_SPLIT_STREAMS=0 tools/probe_run.sh "$ARGS" base | sed "s/^base/base_s0 [$ARGS]/"
    elif [ "$NAME" = base0 ]; then MM2C_COMPACT_RING=0 tools/probe_run.sh "$ARGS" base | sed "s/^base/base0 [$ARGS]/"
    else tools/probe_run.sh "$ARGS" $NAME | sed "s/^$NAME/$NAME [$ARGS]/"; fi
  done
done
