#!/bin/bash
# Build an A/B variant of libpnec_hip.so next to the in-tree one:
#   tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DPNEC_ROW_SWIZZLE>"
# -> pnec_amd/csrc/build/var_<name>/libpnec_hip.so   (use with PNEC_HIP_LIB=... or tools/ab_variants.sh)
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../pnec_amd/csrc"
OUT=build/var_$NAME
mkdir -p $OUT
for f in pnec_capi pnec_frontend pnec_solve_nec pnec_solve_target pnec_solve_host pnec_solve_sym pnec_stream_nec pnec_stream_target pnec_stream_host pnec_stream_sym; do
  [ -f $f.hip ] || continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function \
        -mllvm -amdgpu-sched-strategy=${SCHED:-max-ilp} $FLAGS -c $f.hip -o $OUT/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $OUT/libpnec_hip.so
echo built $OUT/libpnec_hip.so
