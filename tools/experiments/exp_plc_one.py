"""per_layer_chunk_steps of bench.py for the configs[2] and configs[4] shapes only (A/B of host / flush changes): EASYKV_HIP_LIB=... python tools/experiments/exp_plc_one.py"""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda")
for rep in range(2):
    r = bench.per_layer_chunk_steps(types.SimpleNamespace(layers=32, heads=32, kv_heads=8, head_dim=128), dev, 4096, 16, budget=0.3)
    print("c2-shape", round(r["us_per_layer"], 1), round(r["us_per_layer_immediate_scorer"], 1), flush=True)
    r = bench.per_layer_chunk_steps(types.SimpleNamespace(layers=32, heads=32, kv_heads=0, head_dim=128), dev, 10253, 96, n_steps=4, mode="ppl", budget=4096 / 10253,
                                    streaming=True, shape=(40, 40, 40))
    print("c4-shape", round(r["us_per_layer"], 1), round(r["us_per_layer_immediate_scorer"], 1), flush=True)
