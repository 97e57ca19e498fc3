"""per_layer_chunk_steps of bench.py alone (one layer per call, deferred vs immediate scorer) at the five BASELINE shapes."""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
args = types.SimpleNamespace(layers=32, heads=32, kv_heads=0, head_dim=128)
dev = torch.device("cuda")
for a in ((4096, 8), (4096, 64), (9994, 96)):
    r = bench.per_layer_chunk_steps(args, dev, *a)
    print(a, round(r["us_per_layer"], 1), round(r["us_per_layer_immediate_scorer"], 1), round(r["roofline_step"]["frac"], 3), flush=True)
r = bench.per_layer_chunk_steps(args, dev, 10253, 96, n_steps=4, mode="ppl", budget=4096 / 10253, streaming=True, shape=(40, 40, 40))
print("c4", round(r["us_per_layer"], 1), round(r["us_per_layer_immediate_scorer"], 1), round(r["roofline_step"]["frac"], 3))
