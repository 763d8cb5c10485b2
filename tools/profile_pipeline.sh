#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the full-pipeline timing script.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_pipeline_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
export PNEC_NO_INFLIGHT=1   # one call at a time: per-kernel durations of launches that have the GPU to themselves
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pipe -- python $REPO/tools/bench_pipeline.py 20000 > $OUT/run.log 2>&1
grep -E "pnec_hip|Name" $OUT/pipe_kernel_stats.csv | cut -c1-200
find $OUT -type f -size +4M -delete
