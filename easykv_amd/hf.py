"""HF transformers >= 5 seam for the budgeted-KV path.

The reference monkey-patches ``LlamaAttention.forward`` per instance (easykv/utils.py:5-51, easykv/easykv.py:253-256),
which cannot bind to transformers >= 5 (different forward signature, no ``self.rotary_emb``).  The equivalent seam is
``AttentionInterface``: the stock attention module projects and rotates q/k/v, calls ``past_key_values.update()`` (which
here just hands the NEW rows back) and then the registered attention function, which runs the fused HIP step on the
:class:`~easykv_amd.api.BudgetedKVCache` of the forward in flight.  No attention mask is built (the implementation is
not in the mask registry) and no probability matrix is ever returned.

    model = AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.float16).cuda()
    easykv_amd.hf.patch_model(model)
    easykv_amd.enable_fixed_kv(model, tokenizer, mode='auto', stride=8)
"""
from __future__ import annotations

import torch

from . import api

IMPL = "easykv_amd"
_registered = False


def _easykv_attention(module, query, key, value, attention_mask=None, dropout=0.0, scaling=None, **kwargs):
    cache = api.active_cache()
    if cache is None:
        raise RuntimeError("easykv_amd attention called outside easykv_generate(): no BudgetedKVCache is active")
    d = query.shape[-1]
    if scaling is not None and abs(scaling * (d ** 0.5) - 1.0) > 1e-3:
        raise ValueError("easykv_amd attention supports the standard 1/sqrt(head_dim) scaling only")
    if cache.streaming:
        # streaming=True (llama_patch.py:310-327) caches UN-rotated keys; the stock module has already applied RoPE at the
        # true positions, so take it off again (rotation by -theta: x = x'*cos - rotate_half(x')*sin, fp32)
        if cache.unrotate is None or cache.positions is None:
            raise RuntimeError("streaming=True through the HF seam needs easykv_generate() to supply the rotary tables")
        cos, sin = (t[cache.positions].to(torch.float32) for t in cache.unrotate[:2])      # [n, D]
        # the module rotated with cos*s, sin*s (s = attention_scaling: 1 except yarn / longrope): x' = s*R(x).  The kernel
        # rotates with the PURE tables at read time, so the model's s^2 on the logits is put back on the query alone
        s = float(cache.unrotate[2]) if len(cache.unrotate) > 2 else 1.0
        query, key = _unrotate(query, cos, sin, s), _unrotate(key, cos, sin, 1.0 / s)
    out = cache.attend(module.layer_idx, query, key, value)        # [1, Hq, n, D] fp16
    return out.transpose(1, 2).to(query.dtype), None


def _unrotate(x, cos, sin, gain=1.0):
    """Inverse of the PURE rotation (cos, sin) applied to x, times ``gain``."""
    xf = x.to(torch.float32)
    half = xf.shape[-1] // 2
    rot = torch.cat((-xf[..., half:], xf[..., :half]), dim=-1)      # rotate_half (llama_patch.py:13-17)
    out = xf * cos - rot * sin
    return (out * gain if gain != 1.0 else out).to(x.dtype)


def rope_tables_from_model(model, n_pos, head_dim, device):
    """fp32 PURE rotation tables (cos, sin) ``[n_pos, head_dim]`` + the module's ``attention_scaling`` s, from the model's
    own rotary module: cos/sin of ``outer(position, inv_freq)`` in the ``cat(freqs, freqs)`` layout — so rope_theta and
    linear / llama3 / yarn frequency scaling are the model's, while s (yarn, longrope: the module multiplies its tables by
    it) is kept apart: the un-rotation has to divide it out and the logits keep their s^2 through the query.
    DynamicNTK recomputes ``inv_freq`` whenever the sequence outgrows ``max_seq_len_cached``: the tables are only the
    inverse of what the model applied while that does not happen, so the base must be frozen first
    (``set_dynamicntk_rope_length(model, max_len)``, easykv/utils.py:53-57) for sequences that long."""
    for sub in model.modules():
        if hasattr(sub, "inv_freq") and hasattr(sub, "rope_type"):
            if sub.rope_type == "dynamic" and n_pos > int(getattr(sub, "max_seq_len_cached", 0) or 0) + 64:
                raise ValueError(f"streaming=True needs positions up to {n_pos} but the DynamicNTK rotary module would change its "
                                 f"frequencies beyond {sub.max_seq_len_cached}: call set_dynamicntk_rope_length(model, max_length) first")
            if sub.rope_type == "longrope":
                raise ValueError("streaming=True is not supported with longrope (position-dependent frequency sets)")
            inv_freq = sub.inv_freq.to(device=device, dtype=torch.float32)
            if inv_freq.numel() * 2 != head_dim:
                raise ValueError("partial rotary embeddings are not supported by the RoPE-on-read kernels")
            freqs = torch.outer(torch.arange(n_pos, device=device, dtype=torch.float32), inv_freq)
            emb = torch.cat((freqs, freqs), dim=-1)
            return emb.cos().contiguous(), emb.sin().contiguous(), float(getattr(sub, "attention_scaling", 1.0) or 1.0)
    raise ValueError("no rotary embedding module found in this model")


def register():
    global _registered
    if not _registered:
        from transformers import AttentionInterface
        AttentionInterface.register(IMPL, _easykv_attention)
        _registered = True


def patch_model(model):
    """Route every attention layer of a HF Llama/Mistral-style model through the HIP path."""
    register()
    cfg = model.config
    if getattr(cfg, "sliding_window", None) and getattr(cfg, "model_type", "") == "mistral":
        # the reference ignores Mistral's sliding window as well (easykv/mistral_patch.py:90-186)
        pass
    cfg._attn_implementation = IMPL
    for sub in model.modules():
        sub_cfg = getattr(sub, "config", None)
        if sub_cfg is not None and hasattr(sub_cfg, "_attn_implementation"):
            sub_cfg._attn_implementation = IMPL
    return model


class _SkippedLayer(torch.nn.Module):
    """Stands in for a decoder layer another rank owns: the hidden state passes through untouched."""

    def forward(self, hidden_states, *args, **kwargs):
        return hidden_states


def shard_model(model, shard):
    """Layer-shard a stock HF Llama / Mistral decoder stack over the ranks of ``shard`` (easykv_amd.dist.LayerShard): this
    process keeps the decoder layers ``[shard.begin, shard.end)`` — their weights, and through ``easykv_generate`` their K/V
    banks — drops the others, receives the hidden state of every forward from the previous stage in front of its first layer and
    sends it to the next stage behind its last one (one point-to-point transfer per stage boundary and forward, RCCL over xGMI
    with backend ``nccl``).  Embedding, final norm and lm_head stay on every rank; only the last stage's logits are meaningful and
    ``easykv_generate`` samples there and broadcasts the token.

    The reference gets this partition from accelerate's ``device_map='auto'`` (test_passkey.py:25-35, test_ppl.py:25-38), whose
    hooks copy the hidden state between devices — and every layer's probability matrix to one device (llama_patch.py:244-246)."""
    from .dist import PipelineStage
    layers = None
    for sub in model.modules():
        cand = getattr(sub, "layers", None)
        if isinstance(cand, torch.nn.ModuleList) and len(cand) == model.config.num_hidden_layers:
            layers = cand
            break
    if layers is None:
        raise ValueError("no decoder stack (a ModuleList `layers` of config.num_hidden_layers modules) found in this model")
    if shard.n_layers != len(layers):
        raise ValueError(f"shard covers {shard.n_layers} layers, the model has {len(layers)}")
    stage = PipelineStage(shard)
    for l in range(len(layers)):
        if not (shard.begin <= l < shard.end):
            layers[l] = _SkippedLayer()
    if shard.world > 1 and shard.count > 0:
        def recv_hook(module, args, kwargs):
            if stage.first:
                return None
            hidden = args[0] if args else kwargs["hidden_states"]
            got = stage.recv_hidden(hidden)
            return ((got,) + tuple(args[1:]), kwargs) if args else (args, dict(kwargs, hidden_states=got))

        def send_hook(module, args, kwargs, output):
            stage.send_hidden(output[0] if isinstance(output, tuple) else output)
            return output

        layers[shard.begin].register_forward_pre_hook(recv_hook, with_kwargs=True)
        layers[shard.end - 1].register_forward_hook(send_hook, with_kwargs=True)
    model.layer_shard = shard
    return model
