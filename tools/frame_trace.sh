# Runs ON THE GPU BOX: rocprofv3 kernel trace of the per-frame PNEC::Solve demo (host_demo solve_latency); prints the last two frames' kernels (start us, duration us, name).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s2/frame
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/s2/frame -o f -- $R/pnec_amd/pnec_host_demo 512 solve_latency 50 ${FRAME_MODE:-default} ${FRAME_SCHEME:-2} > $R/gpurun_out/s2/frame/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/s2/frame/**/f_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame: find the last occurrence of the first kernel name in a frame
names=[r["Kernel_Name"][:50] for r in rows]
# take the last 14 kernels
last=rows[-14:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("%8.1f %8.1f  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:70]))
PY
tail -2 $R/gpurun_out/s2/frame/run.log | cut -c1-300
