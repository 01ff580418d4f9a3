#!/bin/bash
# Registers, scratch and occupancy of the hot kernels as the compiler reports them (no GPU needed): a change in a big kernel that
# pushes it over a register boundary shows here first (round 3: one `break` in resident_icp_kernel took its scratch from 88 to
# 296 bytes per lane and C3 from 7.45 to 7.9 ms).   usage: tools/kernel_resources.sh [file.hip ...]
cd "$(dirname "$0")/../kss-icp_amd" || exit 1
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
for f in ${@:-csrc/kss_resident.hip csrc/kss_grid.hip csrc/kss_kernels.hip}; do
  /opt/rocm/bin/hipcc $FL -c $f -o /tmp/_kr.o 2> /tmp/_kr.log || { echo "compile failed: $f"; exit 1; }
  python3 - "$f" <<'PY'
import re, sys
name = None; rows = {}
for l in open('/tmp/_kr.log'):
    m = re.search(r'Function Name: (\S+)', l)
    if m: name = m.group(1); rows[name] = {}
    m = re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)', l)
    if m and name: rows[name][m.group(1).split(' [')[0]] = int(m.group(2))
import subprocess
print(sys.argv[1])
for n, r in rows.items():
    if not any(k in n for k in ("resident_icp", "grid_pass_kernel", "gridb_pass", "cand_", "nn_sweep_kernelILi4ELb0ELb0", "rot_search_kernelILi4ELi256", "preshape")): continue
    try: d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split('(')[0]
    except OSError: d = n
    print("  %-80s VGPRs %3d  spilled %3d  scratch %4d B  waves/SIMD %d  LDS %6d" % (d[:80], r.get("VGPRs", -1), r.get("VGPRs Spill", 0), r.get("ScratchSize", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
PY
done
