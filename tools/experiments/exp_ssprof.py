"""Cycle stamps of the generic scorer (EKV_TAIL_PROFILE build): phases per head at a wide-stride shape."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")
def run(S, stride, L=32, Hq=32, H=32, D=128, budget=0.5):
    bp, idx, r_idx = geometry("encoding", S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride)
    out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
    for i in range(8):
        q, k, v = rnd(Hq, stride), rnd(H, stride), rnd(H, stride)
        if i == 7:
            bank._ws.zero_()
        bank.attend(plan, q, k, v, out=out)
    torch.cuda.synchronize()
    w = bank._ws[: bank._ws.numel() // 8 * 8].view(torch.int64).cpu().numpy()
    ok = (w > 10**8) & (w < 10**15)
    idxs = np.nonzero(ok)[0]
    # groups of 7 consecutive stamps at stride 8
    starts = [i for i in idxs if i % 8 == idxs[0] % 8 and ok[i:i + 8].all()]
    st = np.array([w[i:i + 8] for i in starts])
    if len(st) == 0:
        print("no stamps found"); return
    print('   stamp7 - stamp3 (from the start of the selection phase):', (st[:, 7] - st[:, 3]).mean().round(0), flush=True)
    d = np.diff(st[:, :7], axis=1)
    print(f"S={S} stride={stride} T={idx+stride}: {len(st)} heads; mean cycles per phase (0->1 fold, 1->2, 2->3 accumulate, 3->4 select, 4->5 victims/compaction, 5->6 write-back):", d.mean(0).round(0).tolist(), "total", (st[:, 6] - st[:, 0]).mean().round(0), "span of the launch", int(st[:, 6].max() - st[:, 0].min()))
run(9994, 96)
