"""Golden vectors of the reference's sampler front end, ``logits_adapter`` (easykv/easykv.py:115-134): temperature scaling,
top-p mask ``cumsum - p > top_p``, renormalisation, un-sort.  Runs only in the build container (imports /root/reference);
writes ``tests/golden/sampler/logits_adapter.npz`` = inputs + the reference's outputs (data, no source).

    python -m oracle.gen_sampler_golden

TEST INFRASTRUCTURE ONLY (see oracle/easykv_oracle.py header)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "sampler", "logits_adapter.npz")


def main():
    sys.path.insert(0, "/root/reference")
    import easykv.easykv as E
    g = torch.Generator().manual_seed(4242)
    rows = {
        "gauss_v97": torch.randn(3, 97, generator=g) * 3.0,
        "peaked_v512": torch.randn(2, 512, generator=g) * 8.0,
        "flat_v33": torch.zeros(1, 33) + torch.randn(1, 33, generator=g) * 1e-3,
        # exact ties inside the nucleus boundary and a 3-D input (the reference flattens [bsz, l, V], :121-126)
        "ties_v16": torch.tensor([[2.0, 2.0, 1.0, 1.0, 0.0, 0.0, -1.0, -1.0] * 2]),
        "batched_3d": torch.randn(2, 3, 40, generator=g) * 2.0,
    }
    out = {}
    n = 0
    for name, logits in rows.items():
        for temperature in (0.7, 1.0, 1e-6):
            for top_p in (0.3, 0.9, 1.0):
                final, raw = E.logits_adapter(logits.clone(), temperature, top_p)
                key = f"{name}|{temperature}|{top_p}"
                out[key + "|logits"] = logits.numpy()
                out[key + "|final"] = final.numpy()
                out[key + "|raw"] = raw.numpy()
                n += 1
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"{n} (logits, temperature, top_p) cases -> {OUT}")


if __name__ == "__main__":
    main()
