"""Does the step time depend on where K and V sit relative to each other?  One process, several banks whose K / V are views of one buffer."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, engine
dev = torch.device("cuda")
def bench(pad, L=32, Hq=32, H=32, D=128, budget=2048, policy="roco"):
    T = budget + 1
    cap = (T + 63 + 63) // 64 * 64
    n = L * H * cap * D
    if pad is not None:
        buf = torch.empty(2 * n * 2 + pad + 4096, dtype=torch.uint8, device=dev)
        base = (-buf.data_ptr()) % 4096
        k = buf[base:base + 2 * n].view(torch.float16).view(L, H, cap, D)
        v = buf[base + 2 * n + pad: base + 4 * n + pad].view(torch.float16).view(L, H, cap, D)
        orig = torch.empty
        state = {"i": 0}
        def fake_empty(*a, **kw):
            if kw.get("dtype") == torch.float16 and len(a) == 4 and a == (L, H, cap, D):
                state["i"] += 1
                return k
            return orig(*a, **kw)
        torch.empty = fake_empty
        orig_like = torch.empty_like
        torch.empty_like = lambda t, **kw: v if t is k else orig_like(t, **kw)
    try:
        bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    finally:
        if pad is not None:
            torch.empty, torch.empty_like = orig, orig_like
    g = torch.Generator(device=dev).manual_seed(5)
    bank.load_rows(torch.randn(L, H, budget, D, generator=g, device=dev).half(), torch.randn(L, H, budget, D, generator=g, device=dev).half())
    bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=g, device=dev), dim=-1).int()
    bank.state_init(T, 0)
    n_in = 16
    qs = torch.randn(n_in, L, Hq, 1, D, generator=g, device=dev).half(); ks = torch.randn(n_in, L, H, 1, D, generator=g, device=dev).half(); vs = torch.randn(n_in, L, H, 1, D, generator=g, device=dev).half()
    o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev); ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)
    res = []
    for blk in range(3):
        m, t0 = 0, time.perf_counter()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e[0].record()
        while True:
            for _ in range(64):
                bank.attend(plan, qs[m % n_in], ks[m % n_in], vs[m % n_in], out=o, evict_ids=ids); m += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > 0.3: break
        e[1].record(); torch.cuda.synchronize(); res.append(round(e[0].elapsed_time(e[1]) / m * 1e3, 1))
    return res[-1], hex(bank.k.data_ptr() % (1 << 24)), hex((bank.v.data_ptr() - bank.k.data_ptr()) % (1 << 24))
print("separate allocations:", bench(None), flush=True)
for pad in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19):
    print("pad", pad, bench(pad), flush=True)
print("separate allocations:", bench(None), flush=True)
