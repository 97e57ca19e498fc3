"""mode 0 of the wide kernel alone (policy 'full'): configs[3] chunk shape (HBM-fed) and the dense prefix (L2-fed).  EASYKV_HIP_LIB selects the build."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")
tag = os.path.basename(os.environ.get("EASYKV_HIP_LIB", "default"))

def chunk(S, stride, ident, L=32, H=32, D=128, n=10):
    bp, idx, r_idx = geometry("encoding", S, 0.5, stride)
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    bank = KVBank(L, H, H, D, cap=idx + stride, device=dev)
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    if not ident:
        bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    plan = StepPlan(policy="full", phase="prefill", accumulate=False)
    q, k, v = rnd(H, stride), rnd(H, stride), rnd(H, stride)
    out = torch.empty(L, H, stride, D, dtype=torch.float16, device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(n + 3):
        if i == 3: e[0].record()
        bank.attend(plan, q, k, v, out=out)
        bank.n_slots = [idx] * L
    e[1].record(); torch.cuda.synchronize()
    return round(e[0].elapsed_time(e[1]) / n * 1e3, 1)

def prefix(n, L=32, H=32, D=128, reps=3):
    g = torch.Generator(device=dev).manual_seed(2)
    q, k, v = (torch.randn(L, H, n, D, generator=g, device=dev).half() for _ in range(3))
    out = torch.empty(L, H, n, D, dtype=torch.float16, device=dev)
    ms = []
    for _ in range(reps + 1):
        bank = KVBank(L, H, H, D, cap=n + 8, device=dev)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record(); bank.attend(StepPlan(policy="full", phase="prefill", accumulate=False), q, k, v, out=out); e[1].record()
        torch.cuda.synchronize(); ms.append(e[0].elapsed_time(e[1])); del bank
    return round(sum(ms[1:]) / reps, 2)

import time
_t=time.time()
_x=torch.randn(8192,8192,device=dev,dtype=torch.float16)
while time.time()-_t<1.5:
    (_x@_x).sum().item()
print(tag, os.environ.get("EKV_NO_PHYS",""), "identity", chunk(9994, 96, True), "scattered", chunk(9994, 96, False), "identity", chunk(9994, 96, True), "configs[3] chunk mode0 us: scattered", chunk(9994, 96, False), "identity", chunk(9994, 96, True),
      "| stride64 scattered", chunk(4096, 64, False), "| 128 rows S=9994:", chunk(9994, 128, False),
      "| dense prefix 4906 ms:", prefix(4906), flush=True)
