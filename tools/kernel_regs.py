#!/usr/bin/env python3
"""Register / spill report of one HIP source: python tools/kernel_regs.py easykv_amd/csrc/<file>.hip [substring ...] [-Dmacro ...]"""
import re, subprocess, sys, os
src = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
filt = [a for a in sys.argv[2:] if not a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", os.path.basename(src), "-o", "/tmp/_regs.o",
       "-Rpass-analysis=kernel-resource-usage"] + defs
out = subprocess.run(cmd, cwd=os.path.dirname(os.path.abspath(src)), capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs Spill", "VGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "ScratchSize [bytes/lane]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur and key not in rows[cur]:
            rows[cur][key] = int(m.group(1))
if "error" in out and not rows:
    print(out)
for name, r in rows.items():
    if filt and not all(f in name for f in filt):
        continue
    print(f"{name}: vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} occ={r.get('Occupancy [waves/SIMD]')} vspill={r.get('VGPRs Spill')} sspill={r.get('SGPRs Spill')} scratch={r.get('ScratchSize [bytes/lane]')}")
