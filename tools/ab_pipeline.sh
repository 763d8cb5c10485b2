#!/bin/bash
# Runs ON THE GPU BOX: A/B of whole-library variants on the one-call pipeline, interleaved on the same box.
# usage: ab_pipeline.sh <pairs> <variant>...   ("tree" = in-tree library; others = pnec_amd/csrc/build/var_<name>/)
B=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = tree ]; then L=""; else L="pnec_amd/csrc/build/var_$v/libpnec_hip.so"; fi
    echo -n "$v: "; PNEC_HIP_LIB=$L python tools/bench_pipeline.py $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['gpu_ms']; print('one-call %.3f ms | nec %.3f ransac %.3f select %.3f weighted %.3f ls %.3f' % (d['gpu_ms_one_call_pipeline'], g['nec_es (no ransac)'], g['ransac_es'], g['inlier_extraction'], g['weighted_es+scf'], g['ls_refinement']))"
  done
done
