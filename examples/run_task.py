#!/usr/bin/env python
"""The reference's five runner scripts as ONE command over easykv_amd (same modes, strides and generation_config values; the
model path, the prompt template and the data are arguments instead of literals):

    decoding        test_decoding.py:23-48        mode 'decoding', stride 1, budgets 300 / 150 slots, roco, greedy, <= 2048 new tokens
    summarization   test_summarization.py:23-51   mode 'encoding', stride 24, budgets 1.0 / 0.5, roco, keep_attention, T = 0.3
    passkey         test_passkey.py:22-68         mode 'encoding', stride 96, budget 0.5, 6 new tokens, greedy
    passkey_ntk     test_passkey_NTK.py:22-72     same with DynamicNTK RoPE (factor 2, length fixed up front) and stride 24
    ppl             test_ppl.py:22-57             mode 'ppl', stride 96, DynamicNTK, budgets 1.0 / 0.5, recency and roco

    python examples/run_task.py passkey --model /path/to/vicuna-7b-v1.5-16k --jsonl passkey_examples_10k.jsonl
    python examples/run_task.py ppl --model /path/to/Llama-2-13b-hf --text doc.txt
    python examples/run_task.py passkey --random-init tiny            # no checkpoint: random weights, synthetic token ids
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/run_task.py passkey --model ...

Where the reference spreads the model with accelerate's device_map='auto' (test_passkey.py:30), a multi-process launch here gives
every rank a contiguous block of decoder layers (easykv_amd.hf.shard_model).  No checkpoint or tokenizer ships with
this repository (no network): `--random-init` runs the same code path on random weights and synthetic ids, which is what
tests/test_hip_hf_adapter.py::test_example_runners does on the GPU box."""
import argparse
import json
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# prompt templates of the model families the reference's scripts name (test_decoding.py:8-21); '{inst}' is the instruction
TEMPLATES = {
    "plain": "{inst}",
    "vicuna": ("A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, detailed, and "
               "polite answers to the user's questions.\n\nUSER: {inst}\nASSISTANT:"),
    "llama2_chat": "[INST] {inst} [/INST]",
    "alpaca": ("Below is an instruction that describes a task. Write a response that appropriately completes the request.\n\n"
               "### Instruction:\n{inst}\n\n### Response:"),
    "zephyr": "<|user|>\n{inst}</s>\n<|assistant|>\n",
}

SHAPES = {   # --random-init: (hidden, intermediate, layers, heads, kv_heads, vocab)
    "tiny": (128, 256, 3, 4, 2, 512),
    "7b": (4096, 11008, 32, 32, 32, 32000),
    "13b": (5120, 13824, 40, 40, 40, 32000),
    "mistral7b": (4096, 14336, 32, 32, 8, 32000),
}


class SyntheticTokenizer:
    """Stands in for a tokenizer when the weights are random: text -> ids by hashing whitespace-separated words."""
    eos_token_id = -1

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, text, return_tensors="pt"):
        words = (text if isinstance(text, str) else text[0]).split()
        ids = torch.tensor([[zlib.crc32(w.encode()) % self.vocab for w in words]], dtype=torch.long)      # (the same ids on every rank)
        return type("Enc", (), {"input_ids": ids})()

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(i)) for i in ids)


def load(args, dynamic_ntk=None, max_pos=None):
    from transformers import AutoConfig, AutoModelForCausalLM, AutoTokenizer, LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM
    from easykv_amd import hf
    rope = dict(rope_type="dynamic", factor=float(dynamic_ntk)) if dynamic_ntk else None
    if args.model:
        config = AutoConfig.from_pretrained(args.model)
        if rope:      # test_ppl.py:27-29: config.rope_scaling = dict(type="dynamic", factor=2); max_position_embeddings = 4096
            rp = dict(getattr(config, "rope_parameters", None) or {})
            rp.update(rope)
            config.rope_parameters = rp
            config.max_position_embeddings = max_pos or config.max_position_embeddings
        model = AutoModelForCausalLM.from_pretrained(args.model, dtype=torch.float16, config=config).eval()
        tokenizer = AutoTokenizer.from_pretrained(args.model)
    else:
        hidden, inter, layers, heads, kv_heads, vocab = SHAPES[args.random_init]
        kw = dict(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                  num_key_value_heads=kv_heads, max_position_embeddings=max_pos or 16384, attn_implementation="sdpa")
        if rope:
            kw["rope_parameters"] = dict(rope, rope_theta=10000.0)
        torch.manual_seed(0)
        cls, ccls = (MistralForCausalLM, MistralConfig) if args.random_init.startswith("mistral") else (LlamaForCausalLM, LlamaConfig)
        model = cls(ccls(**kw)).half().eval()
        tokenizer = SyntheticTokenizer(vocab)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        from easykv_amd import dist as D
        rank, local, world = D.init()
        torch.cuda.set_device(local)
        model = model.cuda()      # (a checkpoint loader that materialises only the owned layers is DESIGN.md §9's item 6)
        hf.patch_model(model)
        hf.shard_model(model, D.LayerShard(rank, world, model.config.num_hidden_layers))
    else:
        model = hf.patch_model(model.cuda())
    return model, tokenizer


def passkey_prompt(n_filler, key, seed=0):
    """A passkey-retrieval prompt of the usual form (a key hidden at a random depth of repeated filler sentences); the reference
    reads its prompts from passkey_examples_{5k,10k}.jsonl (test_passkey.py:41-44), which --jsonl accepts too."""
    g = torch.Generator().manual_seed(seed)
    filler = "The grass is green. The sky is blue. The sun is yellow. Here we go. There and back again. "
    head = "There is an important info hidden inside a lot of irrelevant text. Find it and memorize them. I will quiz you about the important information there. "
    info = f"The pass key is {key}. Remember it. {key} is the pass key. "
    at = int(torch.randint(0, max(n_filler, 1), (1,), generator=g))
    return head + filler * at + info + filler * (n_filler - at)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("task", choices=["decoding", "summarization", "passkey", "passkey_ntk", "ppl"])
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--model", help="path of a HF Llama / Mistral checkpoint")
    src.add_argument("--random-init", choices=sorted(SHAPES), help="random weights of this shape and a synthetic tokenizer")
    ap.add_argument("--template", choices=sorted(TEMPLATES), default="plain")
    ap.add_argument("--text", help="decoding / summarization: file with the instruction / article; ppl: the document")
    ap.add_argument("--jsonl", help="passkey: records with 'input' and 'target' (the reference's passkey_examples_*.jsonl)")
    ap.add_argument("--kv-policy", default=None)
    ap.add_argument("--budgets", type=float, nargs="*", default=None)
    ap.add_argument("--stride", type=int, default=None)
    ap.add_argument("--max-new-tokens", type=int, default=None)
    ap.add_argument("--filler", type=int, default=400, help="passkey without --jsonl: filler sentences blocks in the synthetic prompt")
    ap.add_argument("--ntk-length", type=int, default=None, help="sequence length the DynamicNTK base is fixed for")
    args = ap.parse_args()
    import easykv_amd
    from easykv_amd import set_dynamicntk_rope_length
    rank0 = int(os.environ.get("RANK", "0")) == 0
    say = print if rank0 else (lambda *a, **k: None)
    read = lambda path, default: open(path).read().strip() if path else default
    template = TEMPLATES[args.template]

    with torch.no_grad():
        if args.task == "decoding":
            model, tok = load(args)
            easykv_amd.enable_fixed_kv(model, tok, mode="decoding", stride=args.stride or 1)
            prompt = template.format(inst=read(args.text, "What are the names of some famous actors that started their careers on Broadway?"))
            for budget in args.budgets or [300, 150]:
                gen = dict(temperature=1e-9, top_p=1.0, max_new_tokens=args.max_new_tokens or 2048, budget=int(budget), kv_policy=args.kv_policy or "roco")
                ids = tok([prompt], return_tensors="pt").input_ids.cuda()
                out = model.easykv_generate(input_ids=ids, generation_config=gen)
                say(f"EasyKV-{gen['kv_policy']}(budget {gen['budget']}): {out}")
        elif args.task == "summarization":
            model, tok = load(args)
            easykv_amd.enable_fixed_kv(model, tok, mode="encoding", stride=args.stride or 24)
            article = read(args.text, passkey_prompt(60, 0)[:4000])
            prompt = template.format(inst="Write a SHORT summary of the following text delimited by triple backticks. Return your response "
                                          f"which covers the key points of the text.\n```{article}```")
            for budget in args.budgets or [1.0, 0.5]:
                gen = dict(temperature=0.3, top_p=1.0, max_new_tokens=args.max_new_tokens or 256, budget=float(budget),
                           kv_policy=args.kv_policy or "roco", keep_attention=True)
                ids = tok([prompt], return_tensors="pt").input_ids.cuda()
                out = model.easykv_generate(input_ids=ids, generation_config=gen)
                say(f"EasyKV-{gen['kv_policy']}({budget * 100:.2f}%): {out}")
        elif args.task in ("passkey", "passkey_ntk"):
            ntk = args.task == "passkey_ntk"
            model, tok = load(args, dynamic_ntk=2 if ntk else None, max_pos=4096 if ntk else None)
            if ntk:
                set_dynamicntk_rope_length(model, args.ntk_length or 5200)      # test_passkey_NTK.py:38
            easykv_amd.enable_fixed_kv(model, tok, mode="encoding", stride=args.stride or (24 if ntk else 96))
            if args.jsonl:
                examples = [json.loads(line) for line in open(args.jsonl)]
            else:
                examples = [dict(input=passkey_prompt(args.filler, 10000 + 7919 * i % 89999, seed=i), target=str(10000 + 7919 * i % 89999)) for i in range(2)]
            postfix = "What is the pass key? The pass key is "
            hits = 0
            for ex in examples:
                ids = tok(template.format(inst=ex["input"] + postfix) if args.template != "plain" else ex["input"] + postfix, return_tensors="pt").input_ids.cuda()
                say("-----------------------------------")
                say("#Tokens of Prompt:", ids.shape[1], "Passkey target:", ex["target"])
                for budget in args.budgets or [0.5]:
                    gen = dict(temperature=1e-9, top_p=1.0, max_new_tokens=args.max_new_tokens or 6, budget=float(budget),
                               kv_policy=args.kv_policy or "roco", keep_attention=False)
                    out = model.easykv_generate(input_ids=ids, generation_config=gen)
                    hits += str(ex["target"]) in out
                    say((f"EasyKV-{gen['kv_policy']}({budget * 100:.2f}%):     [" + postfix + out + "]").replace("\n", "\\n"))
            say(f"retrieved {hits} of {len(examples) * len(args.budgets or [0.5])}")
        else:
            model, tok = load(args, dynamic_ntk=2, max_pos=4096)
            ids = tok(read(args.text, passkey_prompt(args.filler, 0)), return_tensors="pt").input_ids.cuda()
            set_dynamicntk_rope_length(model, args.ntk_length or max(11000, ids.shape[-1] + 1))      # test_ppl.py:36
            easykv_amd.enable_fixed_kv(model, tok, mode="ppl", stride=args.stride or 96)
            say("Input token length:", ids.shape[-1])
            for budget in args.budgets or [1.0, 0.5]:
                for policy in ([args.kv_policy] if args.kv_policy else ["recency", "roco"]):
                    ppl = model.easykv_ppl(input_ids=ids, generation_config=dict(budget=float(budget), kv_policy=policy, keep_attention=False))
                    say(f"EasyKV-{policy}-{budget * 100:.2f}% PPL: {float(ppl):.2f}")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from easykv_amd import dist as D
        D.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
