"""MI355X-native budgeted-KV-cache attention path (drop-in for DRSY/EasyKV's hot path).

Public surface mirrors the reference (easykv/__init__.py:1-2)."""
from .api import BudgetedKVCache, enable_fixed_kv, generate, geometry  # noqa: F401
from .engine import KVBank, StepPlan  # noqa: F401


def set_dynamicntk_rope_length(model, max_length):
    """easykv/utils.py:53-57: fix the DynamicNTK RoPE base for ``max_length`` up front, so that the base does not keep
    changing while the sequence grows (the reference calls ``rotary_emb._set_cos_sin_cache(max_length)`` on every
    ``LlamaAttention`` of HF <= 4.37, whose DynamicNTK module only ever recomputes on growth).

    It sits UPSTREAM of the KV path (keys reach the cache rotated, or un-rotated with ``streaming=True``).  transformers
    >= 5 has one rotary module per model whose ``inv_freq`` is recomputed by ``dynamic_rope_update`` whenever the sequence
    outgrows ``max_seq_len_cached`` (and reset for short sequences); the equivalent here is to install the ``inv_freq`` of
    ``seq_len = max_length`` as both the current and the "original" frequencies and to mark ``max_length`` as the cached and
    the original length (growth beyond ``max_length`` still rescales, as in the reference)."""
    try:
        from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    except Exception as exc:  # pragma: no cover
        raise NotImplementedError("set_dynamicntk_rope_length needs transformers") from exc
    if not hasattr(model, "named_modules"):
        raise NotImplementedError("set_dynamicntk_rope_length needs a HF model (nn.Module with rotary embedding modules)")
    n = 0
    for _, module in model.named_modules():
        if getattr(module, "rope_type", None) != "dynamic" or not hasattr(module, "inv_freq"):
            continue
        inv_freq, scaling = ROPE_INIT_FUNCTIONS["dynamic"](module.config, module.inv_freq.device, seq_len=max_length)
        module.register_buffer("inv_freq", inv_freq, persistent=False)
        module.register_buffer("original_inv_freq", inv_freq.clone(), persistent=False)
        module.attention_scaling = scaling
        module.max_seq_len_cached = max_length
        module.original_max_seq_len = max_length      # no reset to the un-scaled frequencies for short sequences
        n += 1
    if n == 0:
        raise ValueError("no DynamicNTK rotary embedding in this model: config.rope_parameters['rope_type'] must be 'dynamic'")
    print(f"DynamicNTKRoPE max length reset to {max_length}")
