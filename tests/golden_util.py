"""Loading of the committed golden vectors (tests/golden/*.npz, written by oracle/gen_golden.py)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    g = dict(meta=meta,
             streams=(torch.from_numpy(z["qs"]), torch.from_numpy(z["ks"]), torch.from_numpy(z["vs"])),
             kinds=z["evict_kinds"], ids=z["evict_ids"], k=z["evict_k"], ranges=z["evict_ranges"],
             out_lens=z["out_lens"], outputs=torch.from_numpy(z["outputs"]))
    return g


def eos_ids(meta):
    """The EOS ids the reference run of this fixture used ([-1] = never stops early) and the fake model's vocabulary."""
    return meta.get("eos_token_ids", [-1]), meta.get("vocab", 16)


def split_ids(g):
    """-> list over per-head eviction steps of int arrays [L,H,k] (sorted along k)."""
    out, off = [], 0
    for k in g["k"]:
        out.append(g["ids"][..., off:off + int(k)])
        off += int(k)
    return out


def split_outputs(g):
    """-> list over forwards of tensors [L,Hq,n,D]."""
    out, off = [], 0
    for n in g["out_lens"]:
        out.append(g["outputs"][:, :, off:off + int(n)])
        off += int(n)
    return out


def trace_events(trace):
    """Normalise an oracle/product Trace to (kinds, [sorted per-head ids], [ranges])."""
    kinds, ph, rg = [], [], []
    for e in trace.evictions:
        if e["kind"] == "per_head":
            kinds.append(0)
            ph.append(np.sort(np.asarray(e["ids"]).astype(np.int32), axis=-1))
        else:
            kinds.append(1)
            rg.append(tuple(int(x) for x in e["range"]))
    return np.array(kinds, dtype=np.int8), ph, rg


# What the output comparisons of a test session actually needed: written to gpurun_out/stable_fractions.txt by
# tests/conftest.py when the session ends, quoted in DESIGN.md §4.
OUT_STATS = dict(calls=0, elements=0, max_err=0.0, max_err_below_one=0.0, ulp_only=0, ulp_only_max_ref=0.0)


def out_close(a, b, tol=1e-3):
    """The north star's output bar: within 1e-3 of the reference on fp16 outputs — flat, rtol = 0.  Wherever the reference value is
    below 1 in magnitude that is the whole test: |a - b| <= 1e-3, nothing granted.  The path's output dtype IS fp16 (as the reference's
    fp16 configurations'), so for |b| >= 1 the rounding of the result itself is granted on top: |a - b| <= 1e-3 + half an fp16 ulp of b
    (at |o| in [2, 4) half an ulp alone is 9.8e-4 — a bound on the UNROUNDED value no fp16 result can be held to).  How many elements
    ever needed that allowance, and the plain max |a - b|, are counted in OUT_STATS."""
    import torch
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    err = (a - b).abs()
    half_ulp = torch.ldexp(torch.ones_like(b), torch.frexp(b)[1] - 12).clamp_min(2.0 ** -25)     # 0.5 * 2^(floor(log2|b|) - 10)
    allow = torch.where(b.abs() < 1.0, torch.full_like(b, tol), tol + half_ulp)
    ulp_only = (err > tol) & (err <= allow)
    st = OUT_STATS
    st["calls"] += 1
    st["elements"] += err.numel()
    if err.numel():
        st["max_err"] = max(st["max_err"], float(err.max()))
        below = err[b.abs() < 1.0]
        if below.numel():
            st["max_err_below_one"] = max(st["max_err_below_one"], float(below.max()))
        n_ulp = int(ulp_only.sum())
        if n_ulp:
            st["ulp_only"] += n_ulp
            st["ulp_only_max_ref"] = max(st["ulp_only_max_ref"], float(b.abs()[ulp_only].max()))
    return bool((err <= allow).all())
