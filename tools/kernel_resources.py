#!/usr/bin/env python3
"""Registers, scratch and occupancy of the raytrace kernel variants, from hipcc's -Rpass-analysis=kernel-resource-usage.
    python tools/kernel_resources.py [file.hip] [substring filter of the demangled name ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else os.path.join(ROOT, "pyc2ray_amd", "csrc", "raytrace.hip")
filters = [a for a in sys.argv[1:] if not a.endswith(".hip")]
extra = os.environ.get("EXTRA", "").split()
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", *extra,
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    g = lambda pat: int(re.search(pat, b).group(1)) if re.search(pat, b) else -1
    d = d.replace("void asora::", "")
    if filters and not all(f in d for f in filters):
        continue
    v, a, sg = g(r"VGPRs: (\d+)"), g(r"AGPRs: (\d+)"), g(r"SGPRs: (\d+)")
    sc, occ = g(r"ScratchSize \[bytes/lane\]: (\d+)"), g(r"Occupancy \[waves/SIMD\]: (\d+)")
    print(f"{d:110s} VGPR {v:4d} AGPR {a:3d} SGPR {sg:4d} scratch {sc:4d} waves/SIMD {occ}")
