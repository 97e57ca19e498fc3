"""MI355X-native budgeted-KV-cache attention path (drop-in for DRSY/EasyKV's hot path)."""
from .engine import KVBank, StepPlan  # noqa: F401
