#!/bin/bash
# Runs ON THE GPU BOX: A/B of front-stage variants on the same box, interleaved.  usage: ab_front.sh <pairs> <variant>...
# ("tree" = the in-tree library; others = pnec_amd/csrc/build/var_<name>/libpnec_hip.so)
B=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = tree ]; then L=""; else L="pnec_amd/csrc/build/var_$v/libpnec_hip.so"; fi
    echo -n "$v: "; PNEC_HIP_LIB=$L python tools/bench_pipeline.py $B 2>&1 | grep -o "\"gpu_ms\": {[^}]*}"
  done
done
