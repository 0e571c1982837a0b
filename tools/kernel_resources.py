#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, device
only, no GPU needed).   python tools/kernel_resources.py pypbr_amd/csrc/ct_backward.hip [substring filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-slp-vectorize", "-fno-gpu-rdc", "-S",
       "--cuda-device-only", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
name, rows = None, {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("pbr::", "").replace("(KArgs, BArgs)", "").replace("(KArgs)", "").replace("void ", "")
        rows[name] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and name:
            rows[name][key] = int(m.group(1))
for n in sorted(rows):
    if flt in n:
        r = rows[n]
        print(f"{n:90s} vgpr {r.get('vgpr'):4d} agpr {r.get('agpr'):3d} sgpr {r.get('sgpr'):4d} scratch {r.get('scratch'):4d} occ {r.get('occ')} lds {r.get('lds')}")
