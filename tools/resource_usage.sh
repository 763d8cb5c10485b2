#!/bin/bash
# usage: tools_resource_usage.sh file.hip  -> table of kernel resource usage (gfx950)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -mllvm -amdgpu-sched-strategy=max-ilp -c "$1" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    if "error" in l: print(l.strip())
    m=re.search(r"remark:\s+(.*?)\s*\[-Rpass",l)
    if not m: continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        cur={"name":t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
print("%-60s %5s %5s %5s %6s %6s %4s %6s"%("kernel","VGPR","AGPR","SGPR","vSpill","sSpill","occ","LDS"))
for r in rows:
    n=re.sub(r"_ZN8pnec_hip","",r["name"])[:60]
    print("%-60s %5s %5s %5s %6s %6s %4s %6s"%(n,r.get("VGPRs"),r.get("AGPRs"),r.get("TotalSGPRs"),r.get("VGPRs Spill"),r.get("SGPRs Spill"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
