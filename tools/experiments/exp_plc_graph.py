"""per_layer_chunk_steps of bench.py, eager loop against hipGraph replay (configs[1], stride 64 and configs[2] shapes)."""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda")
a7 = types.SimpleNamespace(layers=32, heads=32, kv_heads=0, head_dim=128)
for name, kw in (("c1", dict(S=4096, stride=8)), ("s64", dict(S=4096, stride=64)), ("c3", dict(S=9994, stride=96)),
                 ("c2", dict(S=4096, stride=16, budget=0.3, shape=(32, 32, 8)))):
    r = bench.per_layer_chunk_steps(a7, dev, **kw)
    print(name, "eager", round(r["us_per_layer"], 2), "hipgraph", r["us_per_layer_hipgraph"], "immediate", round(r["us_per_layer_immediate_scorer"], 1), flush=True)
