#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the front-stage kernels during tools/bench_pipeline.py.  usage: pmc_pipeline.sh <tag> [pairs]
TAG=$1; B=${2:-20000}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcpipe_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/bench_pipeline.py $B"
rocprofv3 --kernel-trace --kernel-include-regex "eigensolver|es_batch|sums36" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o b -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "eigensolver|es_batch|sums36" --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
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
        m=sum(v[k][c])/len(v[k][c]); print("  %-22s %14.4g  per-wave %12.1f"%(c,m,m/w if w else 0))
PY
find $OUT -type f -size +4M -delete
