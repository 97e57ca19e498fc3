"""One shape of bench.per_layer_chunk_steps (chunk steps one layer per call, deferred scorer), optionally with forced key-range splits
(NSPLIT) — also the target of a rocprofv3 kernel trace: python tools/experiments/exp_plc_one.py S stride"""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import easykv_amd, bench
ns = int(os.environ.get("NSPLIT", "0"))
if ns:
    Orig = easykv_amd.StepPlan
    easykv_amd.StepPlan = lambda **kw: Orig(n_split=ns, **kw)
args = types.SimpleNamespace(layers=32, heads=32, kv_heads=0, head_dim=128)
S, stride = int(sys.argv[1]), int(sys.argv[2])
r = bench.per_layer_chunk_steps(args, torch.device("cuda"), S, stride, n_steps=12)
print("NSPLIT", ns, S, stride, round(r["us_per_layer"], 1), round(r["us_per_layer_immediate_scorer"], 1), flush=True)
