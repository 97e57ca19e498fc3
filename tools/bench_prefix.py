"""Dense causal prefix alone (bench.dense_prefix without the rest of the bench line): [REPS=3] python tools/bench_prefix.py [tokens] [layers]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easykv_amd import KVBank, StepPlan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4906
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Hq = H = 32
D = 128
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(L, h, n, D, generator=g, device=dev).half() for h in (Hq, H, H))
for rep in range(int(os.environ.get("REPS", 3))):
    bank = KVBank(L, Hq, H, D, cap=n + 8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    bank.attend(StepPlan(policy="full", phase="prefill", accumulate=False), q, k, v)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    fl = 4 * Hq * D * n * n / 2 * L
    print(f"prefix {n} x {L} layers: {dt * 1e3:.2f} ms -> {fl / dt / 1e12:.0f} TFLOP/s")
    del bank
