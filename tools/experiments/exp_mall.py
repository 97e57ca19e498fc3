"""would a one-layer decode launch run faster if its K / V rows came from the memory-side cache (256 MB Infinity Cache)
instead of HBM?  Upper bound of any prefetch scheme: the same one-layer attention launch (policy 'full': attention + in-kernel fold, no
scorer) (a) cycling over 32 layers (1.07 GB: every launch is HBM-fed) and (b) repeated on ONE layer (33.5 MB: cache-fed after the first
call), warmed blocks, HIP events.  Also (c): the cycling loop with layer l + 1's rows touched by a plain-load kernel on a side stream
while layer l runs (torch.sum over the layer's K and V as the toucher)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easykv_amd import KVBank, StepPlan
dev = torch.device("cuda")
L, Hq, H, D, budget = 32, 32, 32, 128, 2048
T = budget + 1
g = torch.Generator(device=dev).manual_seed(5)
bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
bank.load_rows(torch.randn(L, H, budget, D, generator=g, device=dev).half(), torch.randn(L, H, budget, D, generator=g, device=dev).half())
bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=g, device=dev), dim=-1).int()
q = torch.randn(L, Hq, 1, D, generator=g, device=dev).half(); k = torch.randn(L, H, 1, D, generator=g, device=dev).half(); v = torch.randn(L, H, 1, D, generator=g, device=dev).half()
o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev)
plan = StepPlan(policy="full", phase="decode", accumulate=False)
views = [(q[l:l + 1], k[l:l + 1], v[l:l + 1], o[l:l + 1]) for l in range(L)]
print("plan (n_split, fused):", bank.step_plan(plan, 1, 0, 1), flush=True)
side = torch.cuda.Stream(dev)
sink = torch.zeros(2, device=dev)

def block(layer_of, touch=False, min_s=0.4):
    best = None
    for b in range(3):
        n, t0 = 0, time.perf_counter()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        while True:
            for _ in range(64):
                l = layer_of(n)
                if touch:      # rows of the NEXT layer: plain loads on a side stream, concurrent with this layer's launch
                    nl = layer_of(n + 1)
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        sink[0] = bank.k[nl, :, :T].float().amax(); sink[1] = bank.v[nl, :, :T].float().amax()
                q1, k1, v1, o1 = views[l]
                bank.attend(plan, q1, k1, v1, layer_begin=l, out=o1)
                bank.n_slots[l] = budget
                n += 1
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > min_s: break
        e[1].record(); torch.cuda.synchronize()
        best = (round(e[0].elapsed_time(e[1]) / n * 1e3, 2), round((time.perf_counter() - t0) / n * 1e6, 2))
    return best
print("us per one-layer launch (HIP events, host wall):")
print("  (a) cycling over 32 layers (HBM-fed):       ", block(lambda n: n % L), flush=True)
print("  (b) the same layer every time (cache-fed):  ", block(lambda n: 7), flush=True)
print("  (b') two layers in turn (67 MB):            ", block(lambda n: 7 + (n & 1)), flush=True)
print("  (b'') four layers in turn (134 MB):         ", block(lambda n: 4 + (n & 3)), flush=True)
print("  (c) cycling + next layer touched on a side stream:", block(lambda n: n % L, touch=True), flush=True)
