"""CPU-side checks (no GPU): host geometry vs the oracle and the README's known answers, the C-ABI library loads
and exports every symbol include/easykv_hip.h declares, and the product path fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from oracle import easykv_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_geometry_matches_oracle_and_readme():
    from easykv_amd.api import geometry
    for length in (100, 101, 512, 4096, 5144, 9994, 10253):
        for stride in (1, 4, 7, 8, 24, 96):
            for budget in (0.3, 0.5, 40, 2048):
                if isinstance(budget, int) and budget >= length:
                    continue
                assert geometry("encoding", length, budget, stride) == O.geometry_encoding(length, budget, stride)
                assert geometry("ppl", length, budget, stride) == O.geometry_ppl(length, budget, stride)
                if isinstance(budget, int) and stride > 1:
                    assert geometry("auto", length, budget, stride) == O.geometry_auto(length, budget, stride)
    # retained slots printed in the reference's README (README.md:153, :211, :314)
    assert geometry("encoding", 5144, 0.5, 24)[1] == 2576
    assert geometry("encoding", 9994, 0.5, 96)[1] == 5002
    assert geometry("ppl", 10253, 0.5, 96)[1] == 5165


def test_abi_exports_every_declared_symbol():
    from easykv_amd import _build, _lib
    header = open(os.path.join(ROOT, "include", "easykv_hip.h")).read()
    declared = set(re.findall(r"\b(ekv_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    if not os.path.exists(_build.LIB):
        _build.build_lib()
    lib = _lib.load()
    raw = ctypes.CDLL(_build.LIB)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.ekv_abi_version() == 3
    assert b"workspace" in lib.ekv_strerror(-3)
    # argument checking happens before any device access: callable without a GPU
    assert lib.ekv_workspace_bytes(None, None) == 0
    assert lib.ekv_bank_reset(None, None) == -1


def test_struct_layout_matches_header():
    from easykv_amd._lib import Bank, Step
    assert ctypes.sizeof(Bank) == 6 * 8 + 5 * 4 + 4 + 8      # 6 pointers, 5 int32, padding, the optional arrive pointer
    assert ctypes.sizeof(Step) == 18 * 4 + 5 * 4 + 2 * 4
    header = open(os.path.join(ROOT, "include", "easykv_hip.h")).read()
    body = header[header.index("typedef struct ekv_step {"):header.index("} ekv_step;")]
    names = re.findall(r"\b([a-z_0-9]+)\s*[,;]", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert names == [n for n, _ in Step._fields_]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_has_no_cpu_fallback():
    import easykv_amd
    with pytest.raises(Exception) as e:
        easykv_amd.KVBank(1, 4, 4, 32, 64, device="cpu")
    assert "no CPU fallback" in str(e.value)
    # nothing under easykv_amd/ may import the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "easykv_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_set_dynamicntk_rope_length_fixes_the_ntk_base():
    """easykv/utils.py:53-57 on transformers >= 5: after the call the rotary module uses the NTK base of ``max_length``
    for short AND long position ids (no recompute on growth up to max_length, no reset for short sequences)."""
    import contextlib
    import io
    transformers = pytest.importorskip("transformers")
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    import easykv_amd
    d, base, factor, orig, max_length = 16, 10000.0, 2.0, 64, 256
    cfg = LlamaConfig(hidden_size=64, num_attention_heads=4, num_hidden_layers=1, intermediate_size=64, vocab_size=32,
                      max_position_embeddings=orig, rope_parameters=dict(rope_type="dynamic", factor=factor, rope_theta=base))

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.rotary_emb = LlamaRotaryEmbedding(cfg)
    m = Holder()
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        easykv_amd.set_dynamicntk_rope_length(m, max_length)
    assert buf.getvalue().strip() == f"DynamicNTKRoPE max length reset to {max_length}"
    ntk_base = base * ((factor * max_length / orig) - (factor - 1)) ** (d / (d - 2))
    inv = 1.0 / (ntk_base ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    x = torch.zeros(1, 1, d)
    for n in (8, 200, 8):                                 # short, long (< max_length), short again
        pos = torch.arange(n).view(1, -1)
        cos, sin = m.rotary_emb(x, pos)
        ref = torch.outer(torch.arange(n, dtype=torch.float32), inv)
        assert torch.allclose(cos[0, :, : d // 2], ref.cos(), atol=1e-5), n
        assert torch.allclose(sin[0, :, : d // 2], ref.sin(), atol=1e-5), n
    with pytest.raises(ValueError):
        easykv_amd.set_dynamicntk_rope_length(torch.nn.Linear(2, 2), 128)
    with pytest.raises(NotImplementedError):
        easykv_amd.set_dynamicntk_rope_length(object(), 128)


def test_integration_md_stub_structs_match_the_binding():
    """The ctypes stub documented in INTEGRATION.md §3 (executed verbatim on the GPU by tests/test_hip_seam.py) declares
    the same struct layouts and signatures as the product's own binding."""
    import ctypes as C
    import os
    import re
    from easykv_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text[text.index("## 3."):], re.S).group(1)
    code = code.replace('"easykv_amd/csrc/libeasykv_hip.so"', repr(os.path.join(root, "easykv_amd", "csrc", "libeasykv_hip.so")))
    ns = {}
    exec(code.split("# one-time")[0], ns)
    for theirs, ours in ((ns["ekv_bank"], _lib.Bank), (ns["ekv_step"], _lib.Step)):
        assert C.sizeof(theirs) == C.sizeof(ours)
        assert [(n, getattr(theirs, n).offset, getattr(theirs, n).size) for n, _ in theirs._fields_] == \
               [(n, getattr(ours, n).offset, getattr(ours, n).size) for n, _ in ours._fields_]
    lib = _lib.load()
    assert [a for a in ns["lib"].ekv_step_attend.argtypes[2:]] == [a for a in lib.ekv_step_attend.argtypes[2:]]


def test_active_cache_is_context_local():
    """No process-wide 'current cache': the seam reads a context variable that generate() sets around each forward."""
    import threading
    from easykv_amd import api
    assert api.active_cache() is None
    tok = api._ACTIVE.set("mine")
    seen = []
    t = threading.Thread(target=lambda: seen.append(api.active_cache()))
    t.start()
    t.join()
    assert seen == [None] and api.active_cache() == "mine"
    api._ACTIVE.reset(tok)
    assert api.active_cache() is None
