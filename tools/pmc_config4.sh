#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + SQ counters of the multi-hypothesis launch (config 4: 64 x 4096 x 64, tools/ab_config4.py),
# group form (default) and one solve per block (PNEC_SOLVE_GROUPS=0).  usage: tools/pmc_config4.sh <tag>
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_config4_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  export PNEC_SOLVE_GROUPS=$g
  rocprofv3 --kernel-trace --stats --kernel-include-regex "lm_solve" --output-format csv -d $OUT/t$g -o k -- python $REPO/tools/ab_config4.py > $OUT/t$g.log 2>&1
  rocprofv3 --kernel-trace --kernel-include-regex "lm_solve" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a$g -o b -- python $REPO/tools/ab_config4.py > $OUT/a$g.log 2>&1
  rocprofv3 --kernel-trace --kernel-include-regex "lm_solve" --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM FETCH_SIZE --output-format csv -d $OUT/b$g -o b -- python $REPO/tools/ab_config4.py > $OUT/b$g.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for g in (1,0):
    print("== PNEC_SOLVE_GROUPS=%d"%g)
    for f in glob.glob("$OUT/t%d/**/k_kernel_stats.csv"%g, recursive=True):
        for r in csv.DictReader(open(f)):
            if "lm_solve" in r["Name"]: print("  trace: %-70s calls %s avg_us %.1f min_us %.1f"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
    v=collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("a","b"):
        for f in glob.glob("$OUT/%s%d/**/b_counter_collection.csv"%(sub,g), recursive=True):
            for r in csv.DictReader(open(f)):
                v[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in v:
        w=sum(v[k]["SQ_WAVES"])/max(1,len(v[k]["SQ_WAVES"]))
        print("  ",k,"waves per launch",w)
        for c in sorted(v[k]):
            m=sum(v[k][c])/len(v[k][c]); print("     %-22s %14.4g  per-wave %12.1f"%(c,m,m/w if w else 0))
PY
find $OUT -type f -size +4M -delete
