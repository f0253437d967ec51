#!/usr/bin/env python3
"""Kernel resource table of one .hip file: python scripts/kres.py jackal_navigation_amd/csrc/sgm_sweep.hip [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    print("%-70s VGPR %4s AGPR %3s spill %3s SGPR %3s occ %s LDS %6s scratch %s" % (k[-70:], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("TotalSGPRs"),
          r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"), r.get("ScratchSize [bytes/lane]")))
