"""Chunk steps of the logits-in-LDS kernel on slot-indexed vs ordered score rows: equivalence, then warmed timing."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")
def make(slot, S, stride, L, Hq, H, D, budget=0.5, seed=1):
    bp, idx, r_idx = geometry("encoding", S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    bank.use_slot_rows = slot
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride)
    return bank, plan, idx
def check(S=4096, stride=8, L=8, Hq=32, H=32, D=128, steps=40, policy="roco"):
    (a, plan, idx), (b, _, _) = make(False, S, stride, L, Hq, H, D), make(True, S, stride, L, Hq, H, D)
    plan.policy = policy
    g = torch.Generator(device=dev).manual_seed(9)
    bad = used = 0
    for i in range(steps):
        q = torch.randn(L, Hq, stride, D, generator=g, device=dev).half(); k = torch.randn(L, H, stride, D, generator=g, device=dev).half(); v = torch.randn(L, H, stride, D, generator=g, device=dev).half()
        (oa, ia), (ob, ib) = a.attend(plan, q, k, v), b.attend(plan, q, k, v)
        used += int(all(b._slot_rows))
        if not torch.equal(ia, ib):
            bad += int((ia != ib).any(-1).sum())
            if bad < 6: print("step", i, "ids differ in heads", (ia != ib).any(-1).nonzero()[:3].tolist())
        if not torch.allclose(oa.float(), ob.float(), atol=2e-3): print("step", i, "out differs", float((oa.float() - ob.float()).abs().max()))
    n = a.n_slots[0]
    ok_map = torch.equal(a.slot_of_pos, b.slot_of_pos)
    dS = float((a.score_sum - b.score_sum).abs().max()); dQ = float((a.score_sq - b.score_sq).abs().max()); dC = float((a.score_cnt - b.score_cnt).abs().max())
    print(f"check {policy} S={S} stride={stride} L={L} Hq={Hq} H={H}: slot steps {used}/{steps}, heads with different ids {bad}, maps equal {ok_map}, |dS| {dS:.2e} |dQ| {dQ:.2e} |dC| {dC}", flush=True)
def bench(slot, S=4096, stride=8, L=32, Hq=32, H=32, D=128, ids=True):
    bank, plan, idx = make(slot, S, stride, L, Hq, H, D)
    g = torch.Generator(device=dev).manual_seed(5)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    qs, ks, vs = [rnd(Hq, stride) for _ in range(3)], [rnd(H, stride) for _ in range(3)], [rnd(H, stride) for _ in range(3)]
    out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
    res, n = [], 0
    for blk in range(3):
        m, t0 = 0, time.perf_counter()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e[0].record()
        while True:
            for _ in range(8):
                bank.attend(plan, qs[n % 3], ks[n % 3], vs[n % 3], out=out, evict_ids=None if ids else False); n += 1; m += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > 0.4: break
        e[1].record(); torch.cuda.synchronize(); res.append(round(e[0].elapsed_time(e[1]) / m * 1e3, 1))
    return res, all(bank._slot_rows)
if "check" in sys.argv:
    check(); check(policy="h2o_head"); check(policy="tova", steps=20); check(stride=4, steps=30); check(stride=2, Hq=32, H=8, L=32, steps=30)
if "bench" in sys.argv:
    for i in range(2): print(os.path.basename(os.environ.get("EASYKV_HIP_LIB", "new")), "ordered", bench(False), "slot", bench(True), "slot no ids", bench(True, ids=False), flush=True)
