"""Cycle stamps of the scorer tail of the wide column-sum pass (build the m2 translation units with -DEKV_TAIL_PROFILE): phases per head."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")
def run(S, stride, L=32, Hq=32, H=32, D=128, budget=0.5, mode="encoding", streaming=False):
    bp, idx, r_idx = geometry(mode, S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    if streaming:
        from easykv_amd.api import rope_tables
        bank.set_rope(*rope_tables(idx + stride + 64, D))
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, streaming=streaming)
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
    starts = [i for i in idxs if i % 8 == idxs[0] % 8 and ok[i:i + 7].all()]
    st = np.array([w[i:i + 7] for i in starts])
    if len(st) == 0:
        print("no stamps found"); return
    d = np.diff(st, axis=1)
    print(f"S={S} stride={stride} T={idx+stride} L={L} H={H}: {len(st)} heads; mean cycles (0->1 loads + accumulate, 1->2 keys, 2->3 select k1, 3->4 mean keys + select k, "
          f"4->5 cells + scan, 5->6 write-back):", d.mean(0).round(0).tolist(), "total", (st[:, 6] - st[:, 0]).mean().round(0),
          "span of all tails", int(st[:, 6].max() - st[:, 0].min()), flush=True)
sel = sys.argv[1:] or ["c3", "s64", "c2"]
if "c3" in sel: run(9994, 96)
if "s64" in sel: run(4096, 64)
if "c2" in sel: run(4096, 16, H=8, budget=0.3)
if "c4" in sel: run(10253, 96, L=40, Hq=40, H=40, mode="ppl", budget=4096 / 10253, streaming=True)
