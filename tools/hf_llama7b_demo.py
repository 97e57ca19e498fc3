#!/usr/bin/env python
"""End-to-end demo on a Llama2-7B-SHAPED, randomly initialised HF model (no checkpoint is available offline):
strided prefill of a 4096-token prompt to a 2048-slot budget, then budgeted decode (auto mode, roco), through
easykv_amd's HF >= 5 seam — next to HF's own full-cache generate on the same model.  Secondary data point only: the
headline metric is bench.py (the attention/evict path alone).

    python tools/hf_llama7b_demo.py [--layers 32] [--prompt 4096] [--budget 2048] [--stride 8] [--new 64]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Tok:
    eos_token_id = -1

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(map(str, ids))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=4096)
    ap.add_argument("--budget", type=int, default=2048)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    from transformers import LlamaConfig, LlamaForCausalLM
    import easykv_amd
    from easykv_amd import hf

    cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=args.layers,
                      num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=8192, attn_implementation="sdpa")
    torch.manual_seed(0)
    t0 = time.time()
    with torch.device("cuda"):
        model = LlamaForCausalLM(cfg).half().eval()
    print(f"random-init Llama2-7B-shaped model ({args.layers} layers, {sum(p.numel() for p in model.parameters()) / 1e9:.2f} B params) in {time.time() - t0:.1f} s", flush=True)
    ids = torch.randint(0, 32000, (1, args.prompt), device="cuda")
    res = {}

    def timed(fn):
        torch.cuda.synchronize()
        t = time.time()
        fn()
        torch.cuda.synchronize()
        return time.time() - t

    # --- HF baseline: full cache, HF's own generate (sdpa), greedy
    with torch.inference_mode():
        gen = lambda n: model.generate(ids, max_new_tokens=n, do_sample=False, min_new_tokens=n)
        timed(lambda: gen(4))
        a, b = timed(lambda: gen(8)), timed(lambda: gen(8 + args.new))
    res["hf_full_cache"] = dict(decode_tok_s=args.new / (b - a), prefill_plus_8_s=a, kv_slots=args.prompt + args.new)
    print("HF full cache:", res["hf_full_cache"], flush=True)

    # --- budgeted path
    hf.patch_model(model)
    easykv_amd.enable_fixed_kv(model, Tok(), mode="auto", stride=args.stride)

    use_graph = False

    def run(n):
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            model.easykv_generate(input_ids=ids, generation_config=dict(budget=args.budget, kv_policy="roco", max_new_tokens=n,
                                                                        temperature=1.0, eos_token_ids=[-1], eos_poll=16, hipgraph=use_graph))
        return buf.getvalue().strip()

    timed(lambda: run(4))
    a = timed(lambda: run(8))
    line = None
    def last():
        nonlocal line
        line = run(8 + args.new)
    b = timed(last)
    res["easykv_amd_auto_roco"] = dict(decode_tok_s=args.new / (b - a), prefill_plus_8_s=a, printed=line,
                                       budget=args.budget, stride=args.stride)
    print("budgeted path:", res["easykv_amd_auto_roco"], flush=True)

    # --- same, with the steady-state decode step of the whole model replayed as one hipGraph (generation_config['hipgraph'])
    if not args.no_graph:
        use_graph = True
        timed(lambda: run(4))
        a = timed(lambda: run(8))
        b = timed(last)
        res["easykv_amd_auto_roco_hipgraph"] = dict(decode_tok_s=args.new / (b - a), prefill_plus_8_s=a, printed=line,
                                                    budget=args.budget, stride=args.stride)
        print("budgeted path + hipGraph decode step:", res["easykv_amd_auto_roco_hipgraph"], flush=True)
    # --- prefill alone (max_new_tokens = 1): eager loop vs the steady-state chunk forward replayed as one hipGraph (round 6), with the
    #     evicted ids of every forward recorded — they must be identical
    def prefill(graph):
        with contextlib.redirect_stdout(io.StringIO()):
            _, cache = model.easykv_generate(input_ids=ids, generation_config=dict(budget=args.budget, kv_policy="roco", max_new_tokens=1, temperature=1.0,
                                                                                   eos_token_ids=[-1], hipgraph=graph, _record_evictions=True), return_cache=True)
        return cache
    caches, secs = {}, {}
    for graph in (False, True):
        timed(lambda: prefill(graph))
        box = []
        secs[graph] = timed(lambda: box.append(prefill(graph)))
        caches[graph] = box[0]
    ev = {g_: [torch.stack(list(e)).cpu() for e in c.evictions] for g_, c in caches.items()}
    same = len(ev[False]) == len(ev[True]) and all(torch.equal(a, b) for a, b in zip(ev[False], ev[True]))
    res["prefill_only"] = dict(prompt=args.prompt, stride=args.stride, eager_s=round(secs[False], 3), hipgraph_s=round(secs[True], 3),
                               evicting_forwards=len(ev[True]), evicted_ids_identical=bool(same))
    print("prefill alone:", res["prefill_only"], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
