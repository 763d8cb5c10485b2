#!/bin/bash
# Runs ON THE GPU BOX: SQ counters for one bench configuration.  usage: pmc_quick.sh <tag> [bench args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --kernel-include-regex "lm_solve|lm_group" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o b -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "lm_solve|lm_group" --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
v=collections.defaultdict(list)
for f in glob.glob("$OUT/*/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        v[r["Counter_Name"]].append(float(r["Counter_Value"]))
w=sum(v["SQ_WAVES"])/max(1,len(v["SQ_WAVES"]))
print("$TAG", "waves",w)
for k in sorted(v):
    m=sum(v[k])/len(v[k]); print("  %-22s %14.4g  per-wave %10.1f"%(k,m,m/w if w else 0))
PY
grep -h "Missing" $OUT/*.log | cut -c1-300 | head -3
