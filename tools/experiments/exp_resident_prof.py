"""Cycle stamps of the logits-resident chunk step (tools/experiments/build_variant.sh resprof "-DEKR_PROFILE" ekv_attn_resident_d128.hip;
run with EASYKV_HIP_LIB=easykv_amd/csrc/variants/lib_resprof.so): phases per head at the configs[2] shape."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")
def run(S, stride, L=32, Hq=32, H=8, D=128, budget=0.3):
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
    starts = [i for i in idxs if i % 8 == idxs[0] % 8 and ok[i:i + 8].all()]
    st = np.array([w[i:i + 8] for i in starts])
    if len(st) == 0:
        print("no stamps found"); return
    d = np.diff(st, axis=1)
    print(f"S={S} stride={stride} T={idx+stride} L={L} H={H}: {len(st)} heads; mean cycles (prologue, K pass, row maxima, V pass with the exponentials, row sums + column sums + append, "
          f"output, scorer):", d.mean(0).round(0).tolist(), "total", (st[:, 7] - st[:, 0]).mean().round(0),
          "span of the launch", int(st[:, 7].max() - st[:, 0].min()), "min / max start", int(st[:, 0].min() - st[:, 0].min()), int(st[:, 0].max() - st[:, 0].min()), flush=True)
run(4096, 16)
