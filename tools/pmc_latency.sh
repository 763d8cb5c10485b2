#!/bin/bash
# Runs ON THE GPU BOX: memory-latency counters (in-flight levels / instruction counts) of the front-stage
# kernels during tools/bench_pipeline.py.  usage: pmc_latency.sh <tag> [pairs] [kernel regex]
TAG=$1; B=${2:-20000}; RE=${3:-eigensolver}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmclat_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/bench_pipeline.py $B"
rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/a -o b -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc SQ_WAVES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
v=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        v[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in v:
    w=sum(v[k]["SQ_WAVES"])/max(1,len(v[k]["SQ_WAVES"]))
    print(k, "waves", w)
    for c in sorted(v[k]):
        m=sum(v[k][c])/len(v[k][c]); print("  %-26s per-wave %12.1f"%(c,m/w if w else 0))
PY
grep -h "Unable" $OUT/*.log | cut -c1-300 | head -3
find $OUT -type f -size +4M -delete
