"""Builds the HIP shared library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("EASYKV_HIP_LIB") or os.path.join(CSRC, "libeasykv_hip.so")   # (override: profiling builds)
# -ffp-contract=off: the score arithmetic must round like the reference's separate torch ops
# (q/c - (s/c)**2); FMAs that are wanted are written as fmaf()/dot2/MFMA explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
OBJ = os.path.join(CSRC, "obj")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +
            [os.path.join(HERE, "..", "include", "easykv_hip.h")])


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """Compile every .hip to an object (in parallel; unchanged objects are reused) and link the .so."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers())
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in __import__("sys").argv, verbose=True))
