#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
Usage: kres.py f5e-tts_amd/csrc/gemm_bf16_pp.hip [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?)( \[-R|$)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for k, r in rows.items():
    print(f"{k[:70]:70s} vgpr {r.get('VGPRs'):>4s} agpr {r.get('AGPRs'):>3s} sgpr {r.get('TotalSGPRs'):>4s} scratch {r.get('ScratchSize'):>4s} "
          f"vspill {r.get('VGPRs Spill'):>3s} occ {r.get('Occupancy')}")
