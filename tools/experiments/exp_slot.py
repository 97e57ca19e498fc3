"""Slot-indexed vs ordered score rows on the one-launch decode step: equivalence over many steps, then warmed timing."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
dev = torch.device("cuda")
def make(slot, policy, L, Hq, H, D, budget, seed=5):
    T = budget + 1
    g = torch.Generator(device=dev).manual_seed(seed)
    bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    bank.use_slot_rows = slot
    k0 = torch.randn(L, H, budget, D, generator=g, device=dev)
    if os.environ.get("HEAVY"): k0 = k0 * torch.exp(1.2 * torch.randn(L, H, budget, 1, generator=g, device=dev))
    bank.load_rows(k0.half(), torch.randn(L, H, budget, D, generator=g, device=dev).half())
    bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=g, device=dev), dim=-1).int()
    bank.state_init(T, 0)
    return bank
def check(policy="roco", L=4, Hq=32, H=32, D=128, budget=2048, steps=300):
    a, b = make(False, policy, L, Hq, H, D, budget), make(True, policy, L, Hq, H, D, budget)
    g = torch.Generator(device=dev).manual_seed(9)
    plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)
    bad = 0
    for i in range(steps):
        q = torch.randn(L, Hq, 1, D, generator=g, device=dev).half(); k = torch.randn(L, H, 1, D, generator=g, device=dev).half(); v = torch.randn(L, H, 1, D, generator=g, device=dev).half()
        oa, ia = a.attend(plan, q, k, v); ob, ib = b.attend(plan, q, k, v)
        if not torch.equal(ia, ib):
            bad += int((ia != ib).sum())
            if bad < 20: print("step", i, "ids differ at", (ia != ib).nonzero()[:3].tolist(), ia[ia != ib][:3].tolist(), ib[ia != ib][:3].tolist())
        if not torch.allclose(oa.float(), ob.float(), atol=2e-3): print("step", i, "out differs", float((oa.float() - ob.float()).abs().max()))
    assert any(b._slot_rows), "slot rows were never used"
    n = a.n_slots[0]
    ok_map = torch.equal(a.slot_of_pos[:, :, :n], b.slot_of_pos[:, :, :n])
    dS = float((a.score_sum - b.score_sum).abs().max()); dQ = float((a.score_sq - b.score_sq).abs().max()) if policy == "roco" else 0.0
    dC = float((a.score_cnt - b.score_cnt).abs().max()) if policy == "roco" else 0.0
    print(f"check {policy} L={L} Hq={Hq} H={H} budget={budget} steps={steps}: id mismatches {bad}, map equal {ok_map}, max|dS| {dS:.3e} |dQ| {dQ:.3e} |dC| {dC}", flush=True)
def bench(slot, policy="roco", L=32, Hq=32, H=32, D=128, budget=2048, want_ids=True):
    bank = make(slot, policy, L, Hq, H, D, budget)
    g = torch.Generator(device=dev).manual_seed(5)
    n_in = 32
    qs = torch.randn(n_in, L, Hq, 1, D, generator=g, device=dev).half(); ks = torch.randn(n_in, L, H, 1, D, generator=g, device=dev); vs = torch.randn(n_in, L, H, 1, D, generator=g, device=dev).half()
    if os.environ.get("HEAVY"): ks = ks * torch.exp(1.2 * torch.randn(n_in, L, H, 1, 1, generator=g, device=dev))
    ks = ks.half()
    o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev); ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)
    res = []
    for blk in range(4):
        n, t0 = 0, time.perf_counter()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]; e[0].record()
        while True:
            for _ in range(64):
                bank.attend(plan, qs[n % n_in], ks[n % n_in], vs[n % n_in], out=o, evict_ids=ids if want_ids else False); n += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > 0.4: break
        e[1].record(); torch.cuda.synchronize(); res.append(round(e[0].elapsed_time(e[1]) / n * 1e3, 1))
    return res
if "check" in sys.argv:
    check("roco", L=32, steps=200); check("roco", L=8); check("h2o_head", L=8); check("tova", L=8, steps=100); check("roco", L=32, Hq=32, H=8, budget=512, steps=200); check("roco", L=8, budget=300, steps=700)
if "bench" in sys.argv:
    for i in range(3):
        print("ordered roco", bench(False), "slot roco", bench(True), "slot roco, no ids", bench(True, want_ids=False), flush=True)
    print("ordered h2o", bench(False, "h2o_head"), "slot h2o", bench(True, "h2o_head"), flush=True)
if "big" in sys.argv:
    for i in range(2):
        print("budget 4096: ordered roco", bench(False, budget=4096), "slot roco", bench(True, budget=4096), flush=True)
    print("mistral shape (Hq 32, H 8) budget 2048: ordered", bench(False, Hq=32, H=8), "slot", bench(True, Hq=32, H=8), flush=True)
    print("mistral shape budget 4096: ordered", bench(False, Hq=32, H=8, budget=4096), "slot", bench(True, Hq=32, H=8, budget=4096), flush=True)
if "one" in sys.argv:
    for i in range(3): print(os.path.basename(os.environ.get("EASYKV_HIP_LIB", "default")), "slot roco", bench(True), flush=True)

if "heavy" in sys.argv:
    for i in range(3): print("heavy" if os.environ.get("HEAVY") else "iid", "ordered roco", bench(False), "slot roco", bench(True), flush=True)
if "drift" in sys.argv:
    print("heavy" if os.environ.get("HEAVY") else "iid", "ordered h2o", bench(False, "h2o_head"), "ordered roco", bench(False), "slot h2o", bench(True, "h2o_head"), flush=True)
