"""stage_workloads of bench.py alone: what one rank of N = 2 / 4 / 8 runs per step, on one GPU."""
import sys, os, types, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
args = types.SimpleNamespace(layers=32, heads=32, kv_heads=0, head_dim=128)
for r in bench.stage_workloads(args, torch.device("cuda"), 2048, "roco"):
    print(r["layers_in_launch"], r.get("sequences_per_launch"), round(r["us_per_step"], 1), r["plan"], round(r["roofline"]["frac"], 3),
          "| single sequence:", (r.get("single_sequence_launch") or {}).get("us_per_step"), flush=True)
