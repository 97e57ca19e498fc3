"""Chunk-phase timing of one Bench-P shape (bench.strided_prefill): python tools/bench_chunk.py [S] [stride] [n_chunks] [kv_heads]"""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from easykv_amd.engine import KVBank  # noqa: E402

KVBank.default_two_pass = int(os.environ.get("TWO_PASS", "0"))   # 1: force the two-pass chunk kernels, -1: exported logits

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 48
kvh = int(sys.argv[4]) if len(sys.argv) > 4 else 0
args = types.SimpleNamespace(layers=32, heads=32, kv_heads=kvh, head_dim=128, policy=os.environ.get("POLICY", "roco"), identity_layout=False)
# other BASELINE shapes: MODE=ppl BUDGET=0.3995 STREAMING=1 SHAPE=40,40,40 (layers, query heads, KV heads)
shape = tuple(int(x) for x in os.environ["SHAPE"].split(",")) if os.environ.get("SHAPE") else None
r = bench.strided_prefill(args, torch.device("cuda"), n_chunks=n_chunks, S=S, stride=stride, mode=os.environ.get("MODE", "encoding"),
                          budget=float(os.environ.get("BUDGET", "0.5")), streaming=os.environ.get("STREAMING", "0") == "1", shape=shape, pmc=False,
                          prewarm_s=float(os.environ.get("PREWARM_S", "0")))      # (0: counter runs need a known number of steps)
if os.environ.get("BENCH_CHUNK_STEPS_OUT"):      # for bench.live_pmc_step: how many chunk steps this process ran (whole + split form)
    with open(os.environ["BENCH_CHUNK_STEPS_OUT"], "w") as f:
        f.write(str(r["steps_run"]))
print(json.dumps({"workload": r["workload"], "us_per_chunk_step": r["us_per_chunk_step"], "frac_of_hbm_peak": r["roofline"]["frac"],
                  "as_two_launches_us": r["as_two_launches_us"], "value": r["value"], "steps_run": r["steps_run"],
                  "us_per_chunk_step_eager_loop": r["us_per_chunk_step_eager_loop"], "timing": r["roofline"]["timing"]}))
