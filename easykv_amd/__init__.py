"""MI355X-native budgeted-KV-cache attention path (drop-in for DRSY/EasyKV's hot path).

Public surface mirrors the reference (easykv/__init__.py:1-2)."""
from .api import BudgetedKVCache, enable_fixed_kv, generate, geometry  # noqa: F401
from .engine import KVBank, StepPlan  # noqa: F401


def set_dynamicntk_rope_length(model, max_length):
    """easykv/utils.py:53-57 pre-sizes HF <= 4.37's DynamicNTK cos/sin cache so the NTK base stays fixed.
    It sits UPSTREAM of this path (keys reach the cache already rotated) and relies on rotary internals that
    transformers >= 5 no longer has; on HF >= 5 set ``config.rope_parameters`` (``rope_type='dynamic'``,
    ``original_max_position_embeddings``) and ``config.max_position_embeddings = max_length`` instead."""
    cfg = getattr(model, "config", None)
    if cfg is None or not hasattr(cfg, "max_position_embeddings"):
        raise NotImplementedError("set_dynamicntk_rope_length needs a HF config with max_position_embeddings")
    cfg.max_position_embeddings = max_length
    print(f"DynamicNTKRoPE max length reset to {max_length}")
