R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/s2/pmcw; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --kernel-include-regex "weighted_eigensolver|ransac2" --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VMEM --output-format csv -d $OUT -o w -- python $R/tools/bench_pipeline.py 20000 > $OUT/log 2>&1
python3 - <<PY
import csv,glob,collections
v=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/w_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        v[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in v:
    w=sum(v[k]["SQ_WAVES"])/max(1,len(v[k]["SQ_WAVES"]))
    print(k, "waves", w, {c: round(sum(v[k][c])/len(v[k][c])/w,1) for c in v[k]})
PY
