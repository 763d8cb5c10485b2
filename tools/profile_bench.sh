#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes for bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...  (copy the summaries into profiles/).
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex lm_solve --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex lm_solve --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex lm_solve --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o bench -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex lm_solve --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o bench -- $BENCH > $OUT/sq2.log 2>&1
find $OUT -name '*.csv' | head -50
# keep only small files (traces of 100k-block kernels are small; drop anything > 8 MB)
find $OUT -type f -size +8M -delete
