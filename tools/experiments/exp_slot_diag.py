"""At the first step where the slot-indexed and the ordered layout pick different victims: how close was the decision?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
from exp_slot import make, dev
def diag(policy, L, budget, steps, Hq=32, H=32, D=128):
    a, b = make(False, policy, L, Hq, H, D, budget), make(True, policy, L, Hq, H, D, budget)
    g = torch.Generator(device=dev).manual_seed(9)
    plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)
    for i in range(steps):
        q = torch.randn(L, Hq, 1, D, generator=g, device=dev).half(); k = torch.randn(L, H, 1, D, generator=g, device=dev).half(); v = torch.randn(L, H, 1, D, generator=g, device=dev).half()
        S0, Q0, C0 = a.score_sum.clone(), a.score_sq.clone(), a.score_cnt.clone()
        K0, _ = a.ordered_kv()
        oa, ia = a.attend(plan, q, k, v); ob, ib = b.attend(plan, q, k, v)
        if not torch.equal(ia, ib):
            l, h, _ = (ia != ib).nonzero()[0].tolist()
            va, vb = int(ia[l, h, 0]), int(ib[l, h, 0])
            T = budget + 1
            keys = torch.cat([K0[l, h].float(), k[l, h].float()], 0)          # [T, D] in order
            rep = Hq // H
            p = torch.softmax((q[l, h * rep:(h + 1) * rep, 0].float() @ keys.T) / D ** 0.5, -1).mean(0)
            S = S0[l, h, :T].double() + p.double()
            if policy == "roco":
                Qs = Q0[l, h, :T].double() + (p.double() ** 2); c = C0[l, h, :T].double() + 1
                mean = S / c; sd = (Qs / c - mean ** 2).clamp_min(0).sqrt()
                print(f"{policy} step {i} head ({l},{h}): ordered picks {va} (mean {mean[va]:.9e}, sd {sd[va]:.9e}), slot picks {vb} (mean {mean[vb]:.9e}, sd {sd[vb]:.9e})")
                k1 = budget - int(budget * 0.3)
                sdp = sd.clone(); sdp[-10:] = 1e9
                kth = torch.topk(sdp, k1, largest=False).values[-2:]
                print("   sd at ranks k1-1, k1:", [f"{x:.9e}" for x in torch.topk(sdp, k1 + 1, largest=False).values[-2:].tolist()])
            else:
                print(f"{policy} step {i} head ({l},{h}): ordered picks {va} (S {S[va]:.12e}), slot picks {vb} (S {S[vb]:.12e}); rel gap {abs(S[va]-S[vb])/S[va]:.2e}")
            return
    print(policy, "no divergence")
diag("h2o_head", 8, 2048, 300)
diag("roco", 8, 300, 700)
diag("roco", 32, 2048, 200)
