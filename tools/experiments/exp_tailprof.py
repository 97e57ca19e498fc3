"""Cycle stamps of the ordered fused decode kernel's tail on heavy-tailed keys: early vs late state."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
dev = torch.device("cuda")
L, Hq, H, D, budget = 32, 32, 32, 128, 2048
T = budget + 1
g = torch.Generator(device=dev).manual_seed(5)
heavy = bool(os.environ.get("HEAVY"))
bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev); bank.use_slot_rows = False
k0 = torch.randn(L, H, budget, D, generator=g, device=dev)
if heavy: k0 = k0 * torch.exp(1.2 * torch.randn(L, H, budget, 1, generator=g, device=dev))
bank.load_rows(k0.half(), torch.randn(L, H, budget, D, generator=g, device=dev).half())
bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=g, device=dev), dim=-1).int()
bank.state_init(T, 0)
n_in = 16
qs = torch.randn(n_in, L, Hq, 1, D, generator=g, device=dev).half(); ks = torch.randn(n_in, L, H, 1, D, generator=g, device=dev); vs = torch.randn(n_in, L, H, 1, D, generator=g, device=dev).half()
if heavy: ks = ks * torch.exp(1.2 * torch.randn(n_in, L, H, 1, 1, generator=g, device=dev))
ks = ks.half()
o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev); ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=budget)
def stamps():
    bank._ws.zero_()
    bank.attend(plan, qs[0], ks[0], vs[0], out=o, evict_ids=ids); torch.cuda.synchronize()
    w = bank._ws[: bank._ws.numel() // 8 * 8].view(torch.int64).cpu().numpy()
    dbg = w[(w >> 62) == 1]
    fb = (dbg >> 40) & 1; bsel = (dbg >> 20) & 0xFFFFF; inb = dbg & 0xFFFFF
    return len(dbg), "fallbacks", int(fb.sum()), "b_sel hist", np.bincount(np.minimum(bsel, 255), minlength=256)[[0,1,2,253,254,255]].tolist(), "in_bin pct 50/90/99", np.percentile(inb, [50, 90, 99]).tolist()
n = 0
for target in (200, 3000, 9000):
    while n < target:
        bank.attend(plan, qs[n % n_in], ks[n % n_in], vs[n % n_in], out=o, evict_ids=ids); n += 1
    print("heavy" if heavy else "iid", "after", n, "steps: heads, mean cycles [0->1 stream, 1->2 combine, 2->3 softmax+acc, 3->4 victim, 4->5 write-back], tail percentiles 50/90/99:", stamps(), flush=True)
