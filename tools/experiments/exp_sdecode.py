"""Bench-D decode step with streaming=True (RoPE-on-read) vs plain, fused 32-layer launch; warmed."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
from easykv_amd.api import rope_tables
dev = torch.device("cuda")
def run(streaming, L=32, Hq=32, H=32, D=128, budget=2048):
    T = budget + 1
    g = torch.Generator(device=dev).manual_seed(5)
    bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    if streaming: bank.set_rope(*rope_tables(T + 128, D))
    bank.load_rows(torch.randn(L, H, budget, D, generator=g, device=dev).half(), torch.randn(L, H, budget, D, generator=g, device=dev).half())
    bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=g, device=dev), dim=-1).int()
    bank.state_init(T, 0)
    n_in = 32
    qs = torch.randn(n_in, L, Hq, 1, D, generator=g, device=dev).half(); ks = torch.randn(n_in, L, H, 1, D, generator=g, device=dev).half(); vs = torch.randn(n_in, L, H, 1, D, generator=g, device=dev).half()
    o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev); ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=budget, streaming=streaming)
    info = bank.step_plan(plan, 1)
    best = None
    for blk in range(3):
        n, t0 = 0, time.perf_counter()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e[0].record()
        while True:
            for _ in range(64):
                bank.attend(plan, qs[n % n_in], ks[n % n_in], vs[n % n_in], out=o, evict_ids=ids); n += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > 0.4: break
        e[1].record(); torch.cuda.synchronize(); best = e[0].elapsed_time(e[1]) / n * 1e3
    return round(best, 1), info
print(os.path.basename(os.environ.get("EASYKV_HIP_LIB", "default")), "streaming", run(True), "plain", run(False), "mistral-shape streaming", run(True, Hq=32, H=8), flush=True)
print("gqa8 (Hq=64, H=8): streaming", run(True, Hq=64, H=8), "plain", run(False, Hq=64, H=8), flush=True)
