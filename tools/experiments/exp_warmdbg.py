"""Debug driver of the warm-started threshold select: hit / miss counts of the hint on a configs[3]-shaped chunk phase."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")
def run(S, stride, L=32, Hq=32, H=32, D=128, budget=0.5, steps=24):
    bp, idx, r_idx = geometry("encoding", S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride)
    out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
    for i in range(steps):
        bank.attend(plan, rnd(Hq, stride), rnd(H, stride), rnd(H, stride), out=out)
    torch.cuda.synchronize()
    st = bank.slot_state.view(torch.int32).cpu().view(-1, 8)
    h6, h7, h5 = st[:, 6].long(), st[:, 7].long(), st[:, 5].long()
    print(f"S={S} stride={stride} H={H} T={idx+stride}: two-sided hits {int((h7 >> 16).sum())} misses {int((h7 & 0xFFFF).sum())}; one-sided hits {int((h6 >> 16).sum())} misses {int((h6 & 0xFFFF).sum())}; margin exponents {sorted(set(h5.tolist()))}")
run(9994, 96); run(4096, 64); run(4096, 16, H=8, budget=0.3)
