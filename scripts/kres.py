"""Kernel resource usage of one HIP source (VGPR / AGPR / spills / LDS / occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: python scripts/kres.py fuxictr_amd/csrc/x.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
       "-x", "hip", "-c", src, "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    if flt and flt not in dem:
        continue
    print("%-70s VGPR %4s AGPR %4s spill %s/%s LDS %6s occ %s" % (
        dem[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
        r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
