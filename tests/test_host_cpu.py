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
    assert lib.ekv_abi_version() == 8
    assert lib.ekv_rows_to_slots(None, 0, 1, 1, None) == -1 and lib.ekv_rows_to_order(None, 0, 1, 1, None) == -1
    assert lib.ekv_step_info(None, None, None, 0) == -1
    assert b"workspace" in lib.ekv_strerror(-3)
    # argument checking happens before any device access: callable without a GPU
    assert lib.ekv_workspace_bytes(None, None) == 0
    assert lib.ekv_bank_reset(None, None) == -1


def test_struct_layout_matches_header():
    from easykv_amd._lib import Bank, Step
    assert ctypes.sizeof(Bank) == 6 * 8 + 5 * 4 + 4 + 3 * 8      # 6 pointers, 5 int32, padding, the optional arrive / birth / slot_state pointers
    assert ctypes.sizeof(Step) == 18 * 4 + 3 * 4 + 4 * 4 + 6 * 4      # 18 int32, 3 floats, 4 int32, the six row strides of ABI 8
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


def test_sampler_matches_reference_fixture():
    """api.logits_adapter (scatter form) against the reference's logits_adapter (easykv/easykv.py:115-134: temperature, top-p
    mask ``cumsum - p > top_p``, renormalise, un-sort by a second sort + gather), on (logits, temperature, top_p) -> final_prob
    vectors produced by the imported reference (oracle/gen_sampler_golden.py).  Same torch ops on the same values in the same
    order per element, so the bar is bit-exact."""
    import numpy as np
    from easykv_amd.api import logits_adapter
    z = np.load(os.path.join(ROOT, "tests", "golden", "sampler", "logits_adapter.npz"))
    keys = sorted({k.rsplit("|", 1)[0] for k in z.files})
    assert len(keys) == 45
    seen_tp = set()
    for key in keys:
        name, temperature, top_p = key.split("|")
        logits = torch.from_numpy(z[key + "|logits"])
        final, raw = logits_adapter(logits, float(temperature), float(top_p))
        ref_final, ref_raw = torch.from_numpy(z[key + "|final"]), torch.from_numpy(z[key + "|raw"])
        assert final.shape == ref_final.shape
        assert torch.equal(final, ref_final), (key, float((final - ref_final).abs().max()))
        assert torch.equal(raw.reshape(ref_raw.shape), ref_raw), key
        # what the fixture exercises: the nucleus really cuts (zeros) for top_p < 1 and rows still sum to 1
        assert torch.allclose(final.sum(-1), torch.ones(final.shape[:-1]), atol=1e-5)
        if float(top_p) < 1.0 and float(temperature) > 1e-3 and name != "flat_v33":
            assert int((final == 0).sum()) > 0
        seen_tp.add((float(temperature), float(top_p)))
    assert {(0.7, 0.3), (0.7, 0.9), (1.0, 0.3), (1.0, 0.9), (1.0, 1.0)} <= seen_tp


def _fake_bank_step(head_dim=128, hq=32, h=32, cap=2112, n_layers=2, **step):
    from easykv_amd._lib import Bank, Step
    bank = Bank(256, 256, 256, 256, 256, 256, n_layers, hq, h, head_dim, cap, None, 256, 256)     # (never dereferenced by a dry run)
    st = Step()
    st.layer_begin, st.layer_count, st.q_len, st.n_slots, st.score_off = 0, n_layers, 1, 2049, 0
    st.policy, st.accumulate, st.n_evict, st.roco_k1, st.roco_tail, st.range_start = 2, 1, 1, 1434, 10, -1
    st.causal, st.count_add, st.sm_div = 1, 1.0, head_dim ** 0.5
    for k, v in step.items():
        setattr(st, k, v)
    return bank, st


def test_step_check_is_a_dry_run_of_step_attend():
    """ekv_step_check (ABI 4): the argument / shape tests of ekv_step_attend without a launch — callable without a GPU."""
    from easykv_amd import _lib
    lib = _lib.load()
    ok = lambda b, s: lib.ekv_step_check(ctypes.byref(b), ctypes.byref(s))
    assert ok(*_fake_bank_step()) == 0                                        # the north-star decode step
    assert ok(*_fake_bank_step(q_len=8, n_slots=2064, n_evict=8, roco_k1=1847, win_lo=4, win_tail=205)) == 0   # configs[1] chunk
    assert ok(*_fake_bank_step(head_dim=48)) == -2                            # EKV_E_UNSUPPORTED: head_dim not built
    assert ok(*_fake_bank_step(roco_k1=4000)) == -1                           # roco_k1 > W
    assert ok(*_fake_bank_step(n_slots=4000)) == -1                           # T > cap
    assert ok(*_fake_bank_step(policy=4, range_start=-1)) == -1               # range policy without a range
    assert ok(*_fake_bank_step(layer_count=3)) == -1
    # deferred scorer: the flush() shape (phases = 8 over all layers) is checked before the first per-layer call
    assert ok(*_fake_bank_step(defer_layers=2, n_split=8, phases=8)) == 0
    assert ok(*_fake_bank_step(defer_layers=2, n_split=0, phases=8)) == -1    # deferred steps need an explicit split count
    # score rows wider than any scorer's LDS (W > ~39 000 columns) are refused before anything is launched
    assert ok(*_fake_bank_step(cap=60032, n_slots=60000, roco_k1=30000, n_split=-1)) == -2
    assert lib.ekv_step_check(None, None) == -1
    # slot-indexed score rows (ABI 6, EKV_PHASE_SLOT_ROWS = 16): only the one-launch decode step of a shape the layout covers
    slot = dict(n_layers=32, layer_count=32, phases=16, phys_extent=2112)
    assert ok(*_fake_bank_step(**slot)) == 0
    assert ok(*_fake_bank_step(**dict(slot, phases=16 | 32))) == 0             # + "the protected tail is known to be consecutive"
    assert ok(*_fake_bank_step(**dict(slot, score_off=5))) == -2               # a scored window that is not the whole cache
    assert ok(*_fake_bank_step(**dict(slot, policy=1, win_lo=4, win_tail=100))) == -2   # a sink window needs ranks
    assert ok(*_fake_bank_step(**dict(slot, rope_on_read=1))) == -2
    assert ok(*_fake_bank_step(**dict(slot, count_add=1.5))) == -2             # counts must stay integers
    assert ok(*_fake_bank_step(**dict(slot, n_layers=2, layer_count=1))) == -2 # a one-layer launch is split: not the one-launch step
    assert ok(*_fake_bank_step(**dict(slot, q_len=8, n_slots=2064, n_evict=8, roco_k1=1847, win_lo=4, win_tail=205))) == -2   # chunk steps read the ordered layout
    b, s = _fake_bank_step(**slot)
    b.birth = None
    assert ok(b, s) == -2                                                      # a bank without the slot-layout arrays


def test_bench_refuses_to_label_fewer_ranks_as_n_gpus():
    """``python bench.py --gpus N`` without a launcher starts N ranks itself; on a box with fewer GPUs than N it must
    refuse, not run one rank and print ``n_gpus: 1`` (this container has no GPU at all)."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


def test_step_info_launch_counts_of_the_baseline_shapes():
    """ekv_step_info is a dry run (no GPU, no memory touched): the dispatch decisions of the BASELINE shapes, incl. `n_launches` (ABI 7).
    A two-pass wide step with unsplit heads is TWO launches since round 5 (the scorer is the tail of the column-sum pass,
    easykv_amd/csrc/ekv_wide_tail.h; reference easykv/easykv.py:443-499); split heads and RoPE-on-read keep the scorer launch; a step of
    9..64 folded rows against at most 1280 keys is ONE launch since round 6 (easykv_amd/csrc/ekv_attn_resident.inc)."""
    from easykv_amd import _lib
    from easykv_amd._lib import Bank
    from easykv_amd.api import geometry
    from easykv_amd.engine import KVBank, StepPlan

    def bank(L, Hq, H, D, cap, n):
        b = KVBank.__new__(KVBank)
        b.lib = _lib.load()
        cap = (cap + 63) // 64 * 64
        b.n_layers, b.n_q_heads, b.n_kv_heads, b.head_dim, b.cap = L, Hq, H, D, cap
        b._bank = Bank(256, 256, 256, 256, 256, 256, L, Hq, H, D, cap, 256, 256, 256)       # (dummy non-null pointers: nothing is dereferenced)
        b.n_slots, b.extent, b._slot_rows, b._score_sum = [n] * L, [n] * L, [False] * L, True
        return b

    def chunk(L, Hq, H, S, stride, mode="encoding", budget=0.5, streaming=False, lc=None):
        bp, idx, _ = geometry(mode, S, budget, stride)
        plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, streaming=streaming)
        return bank(L, Hq, H, 128, idx + stride, idx).step_info(plan, stride, 0, lc)

    c1 = chunk(32, 32, 32, 4096, 8)
    assert (c1["fused"], c1["n_launches"]) == (1, 1)                                   # configs[1]: the logits-in-LDS kernel
    c2 = chunk(32, 32, 8, 4096, 16, budget=0.3)
    assert (c2["fused"], c2["n_launches"]) == (1, 1), c2                               # configs[2]: the logits-resident kernel (64 rows x 1248 keys)
    for info in (chunk(32, 32, 32, 9994, 96), chunk(16, 32, 32, 9994, 96)):
        assert (info["wide"], info["two_pass"], info["n_split"], info["n_launches"]) == (1, 1, 1, 2), info      # configs[3], a 16-pair stage
    one = chunk(32, 32, 32, 9994, 96, lc=1)
    assert one["n_split"] > 1 and one["n_launches"] == 3                               # one layer per call: split heads keep the scorer launch
    c4 = chunk(40, 40, 40, 10253, 96, "ppl", 4096 / 10253, True)
    assert (c4["wide"], c4["two_pass"], c4["n_launches"]) == (1, 1, 3)                 # configs[4]: RoPE-on-read keeps it too
    b = bank(32, 32, 32, 128, 2049 + 63, 2048)
    dec = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=2048)
    assert (b.step_info(dec, 1)["fused"], b.step_info(dec, 1)["n_launches"]) == (1, 1)
    assert b.step_info(dec, 1, 0, 4)["n_launches"] == 2 and b.step_info(dec, 1, 0, 8)["fused"] == 1      # 128 heads: attention + scorer; 256: the 8-wave fused kernel
    # round 6: launch counts come from the real launch sequence walked dry — phases, the deferred flush, GQA factors the builds pad
    fl = bank(32, 32, 32, 128, 5098 + 63, 5002)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=5093, recent=509, sink=4, stride=96, n_split=4)
    assert fl.step_info(plan, 96, 0, 1)["n_launches"] == 3                            # split heads: one pass + column-sum pass + scorer (which folds)
    assert fl.step_info(plan, 96, 0, 1, phases=1 | 4)["n_launches"] == 3              # one pass + column-sum pass + fold
    assert fl.step_info(plan, 96, 0, 1, phases=1)["n_launches"] == 2                  # attention launches only
    assert fl.step_info(plan, 96, 0, 1, phases=2)["n_launches"] == 1                  # fold + scorer: the stand-alone scorer does both
    assert fl.step_info(plan, 96, 0, 1, phases=8)["n_launches"] == 1
    odd = bank(32, 24, 8, 128, 2049 + 63, 2048)                                        # GQA factor 3 (24 / 8 heads): the REP = 4 build
    assert (odd.step_info(dec, 1)["fused"], odd.step_info(dec, 1)["n_launches"]) == (1, 1)
    wide_gqa = bank(4, 48, 4, 128, 2049 + 63, 2048)                                    # GQA factor 12: query-head groups of 8, generic scorer
    assert wide_gqa.step_info(dec, 1)["fused"] == 0 and wide_gqa.step_info(dec, 1)["n_launches"] == 2      # attention + the generic scorer (folds itself)
    d96 = bank(32, 32, 32, 96, 2049 + 63, 2048)
    assert d96.step_info(dec, 1)["fused"] == 1
    # the dense prefix: 128-row query blocks walked inside one launch (unscored) / one pass + column-sum pass with the tail (scored)
    pre = bank(32, 32, 32, 128, 4906 + 63, 0)
    info = pre.step_info(StepPlan(policy="full", phase="prefill", accumulate=False), 4906)
    assert (info["wide"], info["qb_rows"], info["n_qblocks"], info["n_launches"]) == (1, 128, 39, 1), info
    info = pre.step_info(StepPlan(policy="roco", phase="prefill", accumulate=True), 4906)
    assert (info["wide"], info["two_pass"], info["qb_rows"], info["n_launches"]) == (1, 1, 128, 2), info
