#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
    python tools/kernel_resources.py openset-imagenet_amd/csrc/conv_igemm.hip [substring filter]
The conv kernels live on occupancy: a change that moves a 64x64 kernel past 64 VGPRs or 80 SGPRs costs a resident
workgroup per CU, so this table is checked before and after every kernel edit."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
blocks = re.split(r"remark: Function Name: ", out)[1:]
print(f"{'kernel':80s} SGPR VGPR scratch spillV occ")
for b in blocks:
    name = b.split(" [")[0]
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    if flt and flt not in name:
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [None, "?"])[1]
    print(f"{name[:80]:80s} {g('TotalSGPRs'):>4s} {g('VGPRs'):>4s} {g('ScratchSize [bytes/lane]'):>7s} {g('VGPRs Spill'):>6s} {g('Occupancy [waves/SIMD]'):>3s}")
