#!/bin/bash
# tools/kernel_resources.sh -- per-kernel VGPR/SGPR/scratch/occupancy as the compiler reports them (no GPU needed)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -c \
  -Rpass-analysis=kernel-resource-usage -o /tmp/librtmi_res.o raytracer_amd/csrc/rt_kernels.hip 2>&1 |
python3 -c '
import re,sys,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass",l) or re.search(r"remark:\s+(.*?) \[-Rpass",l)
    if not m: continue
    s=m.group(1).strip()
    if s.startswith("Function Name:") or s.startswith("Name:"):
        cur={"name":s.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in s:
        k,v=s.split(":",1); cur[k.strip()]=v.strip()
names=subprocess.run(["c++filt"]+[r["name"] for r in rows],capture_output=True,text=True).stdout.splitlines()
print("%-44s %5s %5s %7s %5s %6s"%("kernel","VGPR","SGPR","scratch","occ","LDS"))
for r,n in zip(rows,names):
    n=re.sub(r"\(.*","",n)
    print("%-44s %5s %5s %7s %5s %6s"%(n[:44],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
