for ARGS in "--profile dense" "--preset ava-ont" "--profile dense --preset asm20" "--profile mixed" "--preset asm20"; do echo "== $ARGS"; tools/probe_run.sh "$ARGS" base nf1_4 nf1_8 nf4; done
