#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each:
python tools/kernel_resources.py quiver_amd/csrc/qv_batched.hip [name-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if src.endswith("qv_mq64.hip") else []
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", "-o", "/dev/null", src,
                      "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur, d = None, {}
for line in out.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); d[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        d[cur][m.group(1).strip()] = int(m.group(2))
for k, v in sorted(d.items()):
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.split("(")[0].replace("void qv::", "")
    if pat in name:
        print("%-46s vgpr %3d agpr %3d sgpr %3d scratch %4d B  vgpr-spill %3d  lds %6d  waves/SIMD %d" % (
            name[:46], v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize", 0), v.get("VGPRs Spill", 0), v.get("LDS Size", 0), v.get("Occupancy", 0)))
