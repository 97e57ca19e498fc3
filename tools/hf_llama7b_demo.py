#!/usr/bin/env python
"""End-to-end demo on a Llama2-7B-SHAPED, randomly initialised HF model (no checkpoint is available offline):
strided prefill of a 4096-token prompt to a 2048-slot budget, then budgeted decode (auto mode, roco), through
easykv_amd's HF >= 5 seam — next to HF's own full-cache generate on the same model.  Secondary data point only: the
headline metric is bench.py (the attention/evict path alone).

    python tools/hf_llama7b_demo.py [--layers 32] [--prompt 4096] [--budget 2048] [--stride 8] [--new 128] [--only-prefill]

Prefill forms (generation_config extension keys; the default is the reference's forward sequence, one eager forward per chunk):
  hipgraph       the steady-state forwards (evicting chunk, evicting decode step) are captured once and replayed
  dense_growth   the chunks that only grow the cache join the dense prefix as one forward (auto / ppl geometry: half of the forwards)
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


def compare(a, b):
    """Evicted ids of two runs: identical?, first forward that differs (-1: none), share of (forward, layer, head) id sets that are equal."""
    if len(a) != len(b):
        return dict(identical=False, first_diff=-1, share_equal=0.0, forwards=(len(a), len(b)))
    eq = [torch.equal(x, y) for x, y in zip(a, b)]
    share = sum(float((torch.sort(x, -1)[0] == torch.sort(y, -1)[0]).all(-1).float().mean()) for x, y in zip(a, b)) / max(1, len(a))
    return dict(identical=all(eq), first_diff=(eq.index(False) if False in eq else -1), share_equal=round(share, 5))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=4096)
    ap.add_argument("--budget", type=int, default=2048)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--new", type=int, default=128)
    ap.add_argument("--only-prefill", action="store_true", help="skip the HF baseline and the decode runs")
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
    n_par = sum(p.numel() for p in model.parameters()) / 1e9
    print(f"random-init Llama2-7B-shaped model ({args.layers} layers, {n_par:.2f} B params) in {time.time() - t0:.1f} s", flush=True)
    ids = torch.randint(0, 32000, (1, args.prompt), device="cuda")
    res = {}

    def timed(fn):
        torch.cuda.synchronize()
        t = time.time()
        out = fn()
        torch.cuda.synchronize()
        return time.time() - t, out

    # --- HF baseline: full cache, HF's own generate (sdpa), greedy
    if not args.only_prefill:
        with torch.inference_mode():
            gen = lambda n: model.generate(ids, max_new_tokens=n, do_sample=False, min_new_tokens=n)
            timed(lambda: gen(4))
            a, b = timed(lambda: gen(8))[0], timed(lambda: gen(8 + args.new))[0]
        res["hf_full_cache"] = dict(decode_tok_s=round(args.new / (b - a), 1), prefill_plus_8_s=round(a, 3), kv_slots=args.prompt + args.new)
        print("HF full cache:", res["hf_full_cache"], flush=True)

    # --- budgeted path
    hf.patch_model(model)
    easykv_amd.enable_fixed_kv(model, Tok(), mode="auto", stride=args.stride)

    def run(n, record=False, **extra):
        # (temperature 1e-6: the sampled tokens — and with them the evictions of the decode forwards — are the same in every run)
        gc = dict(budget=args.budget, kv_policy="roco", max_new_tokens=n, temperature=1e-6, eos_token_ids=[-1], eos_poll=16,
                  _record_evictions=record, **extra)
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            _, cache = model.easykv_generate(input_ids=ids, return_cache=True, generation_config=gc)
        return buf.getvalue().strip(), cache

    forms = {"eager": {}, "hipgraph": dict(hipgraph=True), "dense_growth": dict(dense_growth=True),
             "hipgraph+dense_growth": dict(hipgraph=True, dense_growth=True)}
    # prefill alone (max_new_tokens = 1), evicted ids of every forward recorded
    evs, pre = {}, {}
    for name, extra in forms.items():
        run(1, **extra)                                     # warm (allocations, kernel attributes, graph pools)
        secs, (line, cache) = timed(lambda: run(1, record=True, **extra))
        evs[name] = [torch.stack(list(e)).cpu() for e in cache.evictions]
        pre[name] = dict(prefill_s=round(secs, 3), printed=line, evicting_forwards=len(evs[name]))
        print(f"prefill alone, {name}:", pre[name], flush=True)
    again = [torch.stack(list(e)).cpu() for e in run(1, record=True)[1].evictions]
    res["prefill_only"] = dict(prompt=args.prompt, stride=args.stride, budget=args.budget, forms=pre,
                               ids_eager_vs_eager_again=compare(evs["eager"], again),
                               ids_eager_vs_hipgraph=compare(evs["eager"], evs["hipgraph"]),
                               ids_eager_vs_dense_growth=compare(evs["eager"], evs["dense_growth"]),
                               ids_dense_growth_vs_hipgraph_dense_growth=compare(evs["dense_growth"], evs["hipgraph+dense_growth"]))
    print("evicted ids:", {k: v for k, v in res["prefill_only"].items() if k.startswith("ids_")}, flush=True)

    # --- decode at the budget: time of `new` more tokens on top of an identical prefill
    if not args.only_prefill:
        for name in ("eager", "hipgraph"):
            extra = forms[name]
            a = timed(lambda: run(8, **extra))[0]
            b, (line, _) = timed(lambda: run(8 + args.new, **extra))
            res[f"decode_{name}"] = dict(decode_tok_s=round(args.new / (b - a), 1), prefill_plus_8_s=round(a, 3), printed=line)
            print(f"budgeted decode, {name}:", res[f"decode_{name}"], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
