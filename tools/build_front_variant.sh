#!/bin/bash
# A/B variant of the front-stage kernels only: recompiles pnec_frontend.hip with extra flags and links it with
# the in-tree objects of everything else.
#   tools/build_front_variant.sh <name> "<extra hipcc flags>"  -> pnec_amd/csrc/build/var_<name>/libpnec_hip.so
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../pnec_amd/csrc"
make -s >/dev/null
OUT=build/var_$NAME
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function \
      -mllvm -amdgpu-sched-strategy=${SCHED:-max-ilp} $FLAGS -c ${SRC:-pnec_frontend.hip} -o $OUT/pnec_frontend.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|Occupancy" | paste - - - - - | sed 's/remark: [^ ]* //g' | cut -c1-250
OBJS=$(ls build/*.o | grep -v pnec_frontend.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/pnec_frontend.o $OBJS -o $OUT/libpnec_hip.so
echo built $OUT/libpnec_hip.so
