#!/usr/bin/env python3
"""VGPR / scratch / spill summary of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py semstereo_amd/csrc/conv3d_bf16s.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
        cur = cur.replace("(anonymous namespace)::", "").replace("void ", "")
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0] + ("_spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
print(f"{'kernel':60s} VGPR AGPR scratch vspill sspill occ")
for k, r in sorted(rows.items()):
    if flt in k:
        print(f"{k[:60]:60s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('ScratchSize', 0):7d} {r.get('VGPRs_spill', 0):6d} {r.get('SGPRs_spill', 0):6d} {r.get('Occupancy', 0):3d}")
