#!/bin/bash
# Runs ON THE GPU BOX: the same command under several builds of the library, interleaved (A B A B ...), one line each.
#   tools/ab_libs.sh "<command>" <reps> <lib or 'default'> [<lib> ...]
CMD=$1; REPS=$2; shift 2
for r in $(seq $REPS); do
  for L in "$@"; do
    if [ "$L" = default ]; then out=$(timeout 300 bash -c "$CMD" 2>/dev/null | tail -1); else out=$(PNEC_HIP_LIB=$L timeout 300 bash -c "$CMD" 2>/dev/null | tail -1); fi
    echo "$L | $out"
  done
done
