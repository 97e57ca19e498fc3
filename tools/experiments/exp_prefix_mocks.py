#!/usr/bin/env python
"""Dense prefix (4906 tokens x 32 layers) on mock builds of the one pass (-DEKW_EXP=10…16: softmax / PV / QK^T left out), one process each."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
names = sys.argv[1:] or ["", "dm10", "dm11", "dm12", "dm13", "dm14", "dm15", "dm16", ""]
for nm in names:
    env = dict(os.environ, REPS="10")
    if nm:
        env["EASYKV_HIP_LIB"] = os.path.join(root, "easykv_amd/csrc/variants", f"lib_{nm}.so")
    out = subprocess.run([sys.executable, os.path.join(root, "tools/bench_prefix.py"), "4906", "32"], env=env, capture_output=True, text=True).stdout
    ms = sorted(float(ln.split(":")[1].split("ms")[0]) for ln in out.splitlines() if ln.startswith("prefix"))
    print(f"{nm or 'shipped':8s} min {ms[0]:.2f} ms  median {ms[len(ms) // 2]:.2f} ms  ({len(ms)} runs)", flush=True)
