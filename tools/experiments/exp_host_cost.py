"""Host cost of one KVBank.attend() call: a chunk step so small that the GPU is never the limit (8 heads x 1 layer, 16 rows x 80 keys),
issued back to back — the per-call interval is what the Python / ctypes path costs."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
dev = torch.device("cuda")
L, Hq, H, D, n, T0 = 1, 32, 8, 128, 4, 64
bank = KVBank(L, Hq, H, D, cap=T0 + n + 64, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
bank.load_rows(rnd(H, T0), rnd(H, T0))
bank.state_init(T0 + n, 2, n)
plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=T0 + n, recent=6, sink=4, stride=n)
q, k, v = rnd(Hq, n), rnd(H, n), rnd(H, n)
out = torch.empty(L, Hq, n, D, dtype=torch.float16, device=dev)
ids = torch.empty(L, H, n, dtype=torch.int32, device=dev)
print(bank.step_info(plan, n))
for rep in range(3):
    for _ in range(200):
        bank.attend(plan, q, k, v, out=out, evict_ids=ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 2000
    for _ in range(N):
        bank.attend(plan, q, k, v, out=out, evict_ids=ids)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue interval {(t1 - t0) / N * 1e6:.1f} us per attend(); incl. drain {(t2 - t0) / N * 1e6:.1f} us", flush=True)
