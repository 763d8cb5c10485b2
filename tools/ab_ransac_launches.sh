#!/bin/bash
# Runs ON THE GPU BOX: the RANSAC stage as one launch (in-tree library) against the two-launch forms of the A/B build
# (tools/build_front_variant.sh twol "-DPNEC_RANSAC_TWO_LAUNCH_AB"): the second launch's list unordered (one bucket) and
# in the order of the rounds still asked for (eight buckets); interleaved, same box.  usage: ab_ransac_launches.sh [pairs] [share]
B=${1:-20000}; OUTL=${2:-0.10}
V=pnec_amd/csrc/build/var_twol/libpnec_hip.so
for rep in 1 2; do
  for s in 2 0; do
    echo -n "one launch:           "; python tools/bench_ransac.py $B 512 $OUTL $s
    echo -n "two, unordered:       "; PNEC_HIP_LIB=$V PNEC_RANSAC_LAUNCHES=2 PNEC_RANSAC_BUCKETS=1 python tools/bench_ransac.py $B 512 $OUTL $s
    echo -n "two, longest first:   "; PNEC_HIP_LIB=$V PNEC_RANSAC_LAUNCHES=2 python tools/bench_ransac.py $B 512 $OUTL $s
  done
done
