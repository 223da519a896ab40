#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel for the .hip sources (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "weather_model_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, f),
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (.*?) \[", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            if cur:
                print(cur)
            mm = re.search(r"\d+([A-Za-z_][A-Za-z_0-9]*kernel)(I[A-Za-z0-9_]*?E)?E", v)
            targs = re.findall(r"L[ib](\d+)E", mm.group(2) or "") if mm else []
            cur = {"kernel": (mm.group(1) if mm else v) + ("<" + ",".join(targs) + ">" if targs else "")}
        else:
            cur[k.replace(" Spill", "Spill").split(" ")[0]] = v
    if cur:
        print(cur)
