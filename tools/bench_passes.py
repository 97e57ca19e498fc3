"""Where does a wide chunk step's time go?  python tools/bench_passes.py [c3 s64 s96 c2 c4 l1]  (EASYKV_HIP_LIB=<variant .so> for A/B builds)
Warmed: every figure is the mean of the LAST of 3 blocks, each block >= 0.3 s of the same launch — a configuration timed cold (first
after process start) runs at lower clocks for hundreds of ms and once cost this repo a whole wrong kernel (docs/TUNING.md §3.5).
 a) mode 0 alone (policy 'full')   b) attention launches of the scored step (phases=1)   c) whole step"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easykv_amd import KVBank, StepPlan, geometry
dev = torch.device("cuda")

def timed(fn, min_s=0.3, blocks=3):
    best = None
    for b in range(blocks):
        n, t0 = 0, time.perf_counter()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        while True:
            for _ in range(8):
                fn(); n += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > min_s: break
        e[1].record(); torch.cuda.synchronize()
        best = e[0].elapsed_time(e[1]) / n * 1e3
    return round(best, 1)

def run(S, stride, L=32, Hq=32, H=32, D=128, mode="encoding", budget=0.5, streaming=False, which=("mode0_full", "attn_launches", "whole_step")):
    bp, idx, r_idx = geometry(mode, S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda h, m: torch.randn(L, h, m, D, generator=g, device=dev).half()
    res = {}
    for name, pol, acc, ev, phases in (("mode0_full", "full", False, False, 0), ("attn_launches", "roco", True, True, 1), ("whole_step", "roco", True, True, 0)):
        if name not in which: continue
        bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
        if streaming:
            from easykv_amd.api import rope_tables
            bank.set_rope(*rope_tables(idx + stride + 64, D))
        bank.load_rows(rnd(H, idx), rnd(H, idx))
        bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
        bank.state_init(idx + stride, 2, stride)
        plan = StepPlan(policy=pol, phase="prefill", accumulate=acc, evict=ev, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, streaming=streaming)
        qs, ks, vs = [rnd(Hq, stride) for _ in range(3)], [rnd(H, stride) for _ in range(3)], [rnd(H, stride) for _ in range(3)]
        out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
        st = {"i": 0}
        def step():
            i = st["i"]; st["i"] += 1
            bank.attend(plan, qs[i % 3], ks[i % 3], vs[i % 3], out=out, phases=phases)
            if pol == "full" or phases == 1:
                bank.n_slots = [idx] * L
        res[name] = timed(step)
        del bank
    print(f"S={S} stride={stride} L={L} H={H} T={idx+stride} streaming={streaming}: {res}", flush=True)

if __name__ == "__main__":
    sel = sys.argv[1:] or ["c3", "s64", "s96", "c2", "c4"]
    if "c3" in sel: run(9994, 96)
    if "s64" in sel: run(4096, 64)
    if "s96" in sel: run(4096, 96)
    if "c2" in sel: run(4096, 16, H=8, budget=0.3)
    if "c4" in sel: run(10253, 96, L=40, Hq=40, H=40, mode="ppl", budget=4096 / 10253, streaming=True)
    if "l1" in sel:
        run(9994, 96, L=1)
        run(4096, 64, L=1)
        run(4096, 16, L=1, H=8, budget=0.3)
        run(10253, 96, L=1, Hq=40, H=40, mode="ppl", budget=4096 / 10253, streaming=True)
