#!/bin/bash
# Runs ON THE GPU BOX: SQ counters (two separate --pmc passes) of the kernels matching a regex during a command.
# usage: tools/pmc_cmd.sh <tag> <kernel-regex> <command...>      (absolute paths in the command: it runs from /tmp)
TAG=$1; RX=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o b -- "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -o b -- "$@" > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
v=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        v[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in v:
    w=sum(v[k]["SQ_WAVES"])/max(1,len(v[k]["SQ_WAVES"]))
    print(k, "waves", w, "launches", len(v[k]["SQ_WAVES"]))
    for c in sorted(v[k]):
        m=sum(v[k][c])/len(v[k][c]); print("  %-22s %14.4g  per-wave %12.1f"%(c,m,m/w if w else 0))
PY
find $OUT -type f -size +4M -delete
