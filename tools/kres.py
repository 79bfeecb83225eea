#!/usr/bin/env python3
"""Register / scratch usage of the kernels in one .hip source of transcodon_amd/csrc (compile only, no GPU):
    python tools/kres.py gemm.hip [name-substring] [extra hipcc flags...]"""
import os, re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "transcodon_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
       "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
if src.startswith("attention"):
    cmd.append("-fno-honor-nans")
out = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
errs = [l for l in out.splitlines() if "error" in l]
print("\n".join(errs[:10]))
cur, rec = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rec[cur] = {}; continue
    m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|TotalSGPRs|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur:
        rec[cur][m.group(1).split()[0]] = int(m.group(2))
for k, v in rec.items():
    if pat in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{name[:90]:90s} vgpr {v.get('VGPRs')} sgpr {v.get('TotalSGPRs')} scratch {v.get('ScratchSize')} occ {v.get('Occupancy')}")
