"""ctypes binding of the C ABI in include/easykv_hip.h.

The product path has NO CPU fallback: if the shared library is missing or a symbol is not
exported this module raises, loudly, at import of the engine.
"""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB

# policy codes (include/easykv_hip.h)
POLICY_NONE, POLICY_H2O_HEAD, POLICY_ROCO, POLICY_TOVA, POLICY_RANGE = 0, 1, 2, 3, 4
PHASE_SLOT_TAIL_OK = 32
PHASE_SLOT_ROWS = 16      # ekv_step.phases bit: the layers' score rows are in the slot-indexed layout (include/easykv_hip.h)
POLICY_CODES = {"full": POLICY_NONE, "h2o_head": POLICY_H2O_HEAD, "roco": POLICY_ROCO, "tova": POLICY_TOVA,
                "recency": POLICY_RANGE, "random": POLICY_RANGE}

EXPORTS = ("ekv_abi_version", "ekv_strerror", "ekv_workspace_bytes", "ekv_step_plan", "ekv_bank_reset", "ekv_state_init",
           "ekv_step_attend", "ekv_gather_ordered", "ekv_scatter_rows", "ekv_compact_inplace", "ekv_step_check", "ekv_step_info",
           "ekv_rows_to_slots", "ekv_rows_to_order")


class Bank(C.Structure):
    _fields_ = [("k", C.c_void_p), ("v", C.c_void_p), ("slot_of_pos", C.c_void_p),
                ("score_sum", C.c_void_p), ("score_sq", C.c_void_p), ("score_cnt", C.c_void_p),
                ("n_layers", C.c_int32), ("n_q_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("cap", C.c_int32), ("arrive", C.c_void_p), ("birth", C.c_void_p), ("slot_state", C.c_void_p)]


class Step(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "layer_begin", "layer_count", "q_len", "n_slots", "score_off", "policy", "accumulate", "n_evict",
        "win_lo", "win_tail", "roco_k1", "roco_tail", "range_start", "tova_head_mean", "causal", "rope_on_read",
        "n_split", "phases")] + [(n, C.c_float) for n in ("count_add", "count_tail_step", "sm_div")] + [(n, C.c_int32) for n in (
        "two_pass", "phys_extent", "defer_layers", "defer_index",
        "q_token_stride", "q_head_stride", "kv_token_stride", "kv_head_stride", "out_token_stride", "out_head_stride")]


class EkvError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and type the shared library.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise EkvError(f"{LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise EkvError(f"{LIB} does not export {name}")
    vp, i32 = C.c_void_p, C.c_int32
    lib.ekv_abi_version.restype = C.c_int
    lib.ekv_strerror.restype = C.c_char_p
    lib.ekv_strerror.argtypes = [C.c_int]
    lib.ekv_workspace_bytes.restype = C.c_size_t
    lib.ekv_workspace_bytes.argtypes = [C.POINTER(Bank), C.POINTER(Step)]
    lib.ekv_step_plan.argtypes = [C.POINTER(Bank), C.POINTER(Step), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ekv_bank_reset.argtypes = [C.POINTER(Bank), vp]
    lib.ekv_state_init.argtypes = [C.POINTER(Bank), i32, i32, i32, i32, i32, vp]
    lib.ekv_step_attend.argtypes = [C.POINTER(Bank), C.POINTER(Step), vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.ekv_gather_ordered.argtypes = [C.POINTER(Bank), i32, i32, i32, vp, vp, vp]
    lib.ekv_scatter_rows.argtypes = [C.POINTER(Bank), i32, i32, i32, i32, vp, vp, vp]
    lib.ekv_compact_inplace.argtypes = [C.POINTER(Bank), i32, i32, i32, i32, vp, vp]
    lib.ekv_step_check.argtypes = [C.POINTER(Bank), C.POINTER(Step)]
    lib.ekv_step_info.argtypes = [C.POINTER(Bank), C.POINTER(Step), C.POINTER(C.c_int32), C.c_int32]
    lib.ekv_rows_to_slots.argtypes = [C.POINTER(Bank), i32, i32, i32, vp]
    lib.ekv_rows_to_order.argtypes = [C.POINTER(Bank), i32, i32, i32, vp]
    for name in EXPORTS[3:]:
        getattr(lib, name).restype = C.c_int
    if lib.ekv_abi_version() != 8:
        raise EkvError("ABI version mismatch")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().ekv_strerror(code).decode()
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        raise EkvError(f"{what}: {msg} ({code})")
