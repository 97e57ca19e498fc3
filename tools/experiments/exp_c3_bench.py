"""Is bench.strided_prefill's configs[3] figure (16 timed steps behind 8 warm-up steps, distinct inputs per step) the steady state?
The same function with longer warm-up / more steps, and exp_passes-style input reuse for comparison."""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
args = types.SimpleNamespace(layers=32, heads=32, kv_heads=0, head_dim=128, policy="roco", identity_layout=False)
dev = torch.device("cuda")
for (n, w) in ((16, 8), (16, 8), (48, 24), (96, 48)):
    r = bench.strided_prefill(args, dev, S=9994, stride=96, n_chunks=n, warm=w, pmc=False)
    print(f"n_chunks={n} warm={w}: {r['us_per_chunk_step']:.1f} us per step, two-launch breakdown {r['as_two_launches_us']}", flush=True)
