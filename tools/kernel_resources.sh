#!/bin/bash
# VGPRs / scratch / LDS of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
# usage: tools/kernel_resources.sh clip-based-cross-modal-hashing_amd/csrc/gemm_wide.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 |
python3 -c '
import re, sys
name = None
rows = {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m: name = m.group(1); rows[name] = {}
    for key in ("VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "LDS Size \[bytes/block\]", "VGPR Spill", "Occupancy \[waves/SIMD\]"):
        m = re.search(r"remark:\s+" + key + r": (\d+)", ln)
        if m and name: rows[name][key.split(" ")[0]] = int(m.group(1))
import subprocess
for n, r in rows.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    print(f"{d[:90]:90s} " + " ".join(f"{k}={v}" for k, v in r.items()))
'
