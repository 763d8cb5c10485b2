#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of a command, per-kernel summary to stdout.
# usage: tools/prof_kernels.sh <outdir-tag> <command...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- "$@" > $OUT/cmd.log 2>&1
F=$(find $OUT -name 'k_kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("%-70s %6s %12s %12s %12s"%("kernel","calls","avg_us","min_us","max_us"))
for r in rows[:25]:
    print("%-70s %6s %12.1f %12.1f %12.1f"%(r["Name"][:70],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
find $OUT -type f -size +8M -delete
