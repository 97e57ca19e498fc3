"""BASELINE.json configs[2] (Mistral GQA, stride 16, budget 0.3, keep_attention) and configs[4] (Llama2-13B head
count, ppl mode, streaming RoPE-on-read, stride 96) END TO END at their full geometry: ``easykv_amd.generate`` over the HIP
engine against the CPU oracle's ``generate`` on the same q/k/v streams (a few layers; the per-layer work is independent).

Same stability rule as tests/test_hip_fullsize.py: a (layer, head) is compared until its first decision that the oracle
itself flips under +-2e-5 relative perturbations of the score rows."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests.golden_util import out_close

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-3   # north_star: within 1e-3 on fp16 outputs — flat absolute bound, rtol = 0


class _ProbeLog:
    PERT = 2e-5

    def __init__(self):
        self.gen = torch.Generator().manual_seed(11)
        self.unstable = []
        self.inputs = []       # per selection: what it takes to re-run the UNSTABLE heads' decision (round 5: see in_tolerance_class)

    def _alt(self, fn, policy, s, q, c, args, dim):
        e1 = 1.0 + (torch.rand(s.shape, generator=self.gen) * 2 - 1) * self.PERT
        e2 = 1.0 + (torch.rand(s.shape, generator=self.gen) * 2 - 1) * self.PERT
        alt = fn(policy, s * e1, q * e2, c.clone(), *args)
        return alt.unsqueeze(-1) if alt.dim() < dim else alt

    def __call__(self, fn, policy, s, q, c, args, ids):
        base = torch.sort(ids, dim=-1)[0]
        bad = torch.zeros(ids.shape[:-1], dtype=torch.bool)
        for _ in range(2):
            bad |= (torch.sort(self._alt(fn, policy, s, q, c, args, ids.dim()), dim=-1)[0] != base).any(dim=-1)
        self.unstable.append(bad)
        flat = lambda x: x.reshape(-1, x.shape[-1]).clone()
        self.inputs.append(dict(fn=fn, policy=policy, args=args, s=flat(s), q=flat(q), c=flat(c), dim=ids.dim()))

    def in_tolerance_class(self, j, flat_row, got_sorted, tries=96):
        """Is ``got_sorted`` one of the ORACLE'S OWN answers for that head under a +-2e-5 perturbation of its scores?"""
        x = self.inputs[j]
        i = flat_row
        for _ in range(tries):
            alt = torch.sort(self._alt(x["fn"], x["policy"], x["s"][i:i + 1], x["q"][i:i + 1], x["c"][i:i + 1], x["args"], 2), dim=-1)[0][0]
            if torch.equal(alt, got_sorted):
                return True
        return False


def _report_stable(what, n_stable, n_dec, frac, floor):
    """The achieved fraction of (layer, head, eviction) decisions that were BOUND to the oracle's: printed (pytest -s / -rA) and
    appended to gpurun_out/stable_fractions.txt so a round's evidence run records it."""
    import os
    line = f"[stable-fraction] {what}: {n_stable}/{n_dec} = {frac:.4f} (floor {floor})"
    print(line)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "stable_fractions.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


# Floor of the BOUND fraction (0.95 for every config).  Round 4 dropped a head at its first UNSTABLE draw (one the oracle itself flips
# under +-2e-5: about one 96-victim draw of 25 at ~5000 columns), which bound 93 % of configs[3] / [4].  Round 5: a head leaves only
# when the two sides actually DECIDED differently on such a draw — and that decision must still be one of the oracle's own answers
# under +-2e-5 (in_tolerance_class); unstable draws both sides resolve alike stay bound.  tests/test_hip_lockstep.py re-seeds the
# oracle from the bank every step instead and binds 100 % of the decisions at the same geometries.
def _run_pair(mode, stride, cfg, n_layers, hq, h, d, length, seed, arch="LlamaForCausalLM", min_stable=0.95):
    import easykv_amd
    from oracle import easykv_oracle as O
    from oracle.fake_model import FakeAttnModel, make_streams
    from tests.native_fake_model import NativeFakeModel
    streams = make_streams(n_layers, hq, h, d, length + cfg.get("max_new_tokens", 0) + 8, seed)
    ids = torch.arange(length).view(1, -1) % 16
    cfg = dict(cfg, eos_token_ids=[-1])

    probe = _ProbeLog()
    O.SELECT_HOOK = probe
    try:
        ref_model = FakeAttnModel(*streams, arch=arch, streaming=cfg.get("streaming", False))
        buf_ref = io.StringIO()
        with contextlib.redirect_stdout(buf_ref):
            tr = O.generate(ref_model, ids, cfg, kv_mode=mode, stride=stride)
    finally:
        O.SELECT_HOOK = None

    model = NativeFakeModel(*streams, arch=arch)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res, cache = easykv_amd.generate(model, ids, dict(cfg, _record_evictions=True), kv_mode=mode, stride=stride, return_cache=True)
    assert buf.getvalue().strip() == tr.report.strip()
    assert cache.bank.n_slots == [tr.cache_len] * n_layers

    per_head = [e for e in tr.evictions if e["kind"] == "per_head"]
    assert len(per_head) == len(probe.unstable) and len(cache.evictions) == len(tr.evictions)
    alive = torch.ones(n_layers, h, dtype=torch.bool)
    n_dec = n_stable = n_class = 0
    j = 0
    for step, (ev, ours) in enumerate(zip(tr.evictions, cache.evictions)):
        got = torch.sort(torch.stack(ours).cpu().long(), dim=-1)[0]
        if ev["kind"] == "range":
            lo, hi = ev["range"]
            assert bool((got == torch.arange(lo, hi)).all()), step
            continue
        ref = torch.sort(ev["ids"].long(), dim=-1)[0]
        ok = ~probe.unstable[j].reshape(got.shape[:-1])
        same = (got == ref).all(dim=-1)
        n_dec += int(alive.sum())
        n_stable += int((alive & (ok | same)).sum())      # bound: well defined and equal, or an unstable draw both sides resolved alike
        # round 5: a decision that came out DIFFERENTLY must still be one of the oracle's own answers under +-2e-5 (whether or not the
        # two probe draws flagged it) — the head then leaves the comparison (its cache differs from here on); an unstable draw that came
        # out the same keeps being compared
        for l, hh in (alive & ~same).nonzero().tolist():
            assert probe.in_tolerance_class(j, l * h + hh, got[l, hh]), f"eviction {step} layer {l} head {hh}: outside the oracle's tolerance class"
            n_class += 1
        j += 1
        alive &= same
    frac = float(n_stable) / max(n_dec, 1)
    _report_stable(f"{mode} stride={stride} L={n_layers} Hq={hq} H={h} S={length} {cfg.get('kv_policy')}"
                   f"{' streaming' if cfg.get('streaming') else ''}{' keep_attention' if cfg.get('keep_attention') else ''}", n_stable, n_dec, frac, min_stable)
    assert n_dec > 0 and frac >= min_stable, (n_stable, n_dec, frac, min_stable)

    # attention outputs of EVERY forward for the query heads whose KV head never left the oracle's trajectory (round 4 compared the
    # first two forwards only once any head had diverged); the dense prefix and the first chunk for all heads
    assert len(model.outputs_log) == len(ref_model.outputs_log)
    keep_q = alive.repeat_interleave(hq // h, dim=1)          # [layers, Hq]
    for f in range(len(model.outputs_log)):
        a, b = model.outputs_log[f], ref_model.outputs_log[f]
        if f < 2:
            assert out_close(a, b, OUT_TOL), (f, float((a - b).abs().max()))
        else:
            assert out_close(a[keep_q], b[keep_q], OUT_TOL), (f, float((a[keep_q] - b[keep_q]).abs().max()))
    return res, tr, frac


def test_config2_mistral_gqa_stride16_keep_attention_full_geometry():
    """configs[2]: S=4096, stride 16, budget 0.3 -> budget'=1244, idx=1232, r_idx=1216, Hq=32 over H=8 KV
    heads, D=128, keep_attention=True: the dense 1216-token prefix also feeds the score rows (easykv/easykv.py:173-186),
    then 180 chunk steps over a cache oscillating 1232 <-> 1248, then a few plain decode steps."""
    from easykv_amd import geometry
    assert geometry("encoding", 4096, 0.3, 16) == (1244, 1232, 1216)
    cfg = dict(budget=0.3, kv_policy="h2o_head", keep_attention=True, max_new_tokens=3, temp_length=4, recent_ratio=0.1)
    res, tr, frac = _run_pair("encoding", 16, cfg, 2, 32, 8, 128, 4096, seed=4242, arch="MistralForCausalLM")
    assert tr.cache_len == 1232 + 3
    assert res == " ".join(str(t) for t in tr.result)


def test_config2_h2o_alias_is_the_references_noop():
    """BASELINE.json writes kv_policy='h2o'; the reference only knows 'h2o_head' (easykv/easykv.py:443-499), any other
    string evicts nothing — reproduced, not 'fixed'."""
    import easykv_amd
    from oracle.fake_model import make_streams
    from tests.native_fake_model import NativeFakeModel
    streams = make_streams(1, 8, 2, 64, 200, 3)
    model = NativeFakeModel(*streams, arch="MistralForCausalLM")
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        _, cache = easykv_amd.generate(model, torch.arange(160).view(1, -1) % 16,
                                       dict(budget=0.3, kv_policy="h2o", max_new_tokens=2, eos_token_ids=[-1]),
                                       kv_mode="encoding", stride=16, return_cache=True)
    assert cache.bank.n_slots == [162]
    assert "(160/160)" in buf.getvalue()
    with contextlib.redirect_stdout(io.StringIO()):
        _, cache = easykv_amd.generate(model, torch.arange(160).view(1, -1) % 16, dict(budget=0.3, kv_policy="h2o", eos_token_ids=[-1]),
                                       kv_mode="ppl", stride=16, return_cache=True)
    assert cache.bank.n_slots == [160]


def test_config4_llama13b_heads_ppl_streaming_stride96_full_geometry():
    """configs[4]: S=10253, stride 96, ppl mode, streaming=True (keys cached un-rotated, RoPE by slot index on every
    read, easykv/llama_patch.py:310-327), 40 heads of D=128.  The reference's ppl mode only evicts for float budgets
    (:759-765), so the run uses the ratio 4096/10253 -> budget'=4192, idx=4109, r_idx=77, W=4205."""
    from easykv_amd import geometry
    ratio = 4096 / 10253
    assert geometry("ppl", 10253, ratio, 96) == (4192, 4109, 77)
    cfg = dict(budget=ratio, kv_policy="roco", streaming=True, temp_length=4, recent_ratio=0.1)
    res, tr, frac = _run_pair("ppl", 96, cfg, 1, 40, 40, 128, 10253, seed=1313)
    assert tr.cache_len == 4109
    assert abs(res - float(tr.result)) <= 1e-6 * abs(float(tr.result))


def test_config3_vicuna_passkey_stride96_full_geometry():
    """configs[3]: Vicuna-7B-16K passkey retrieval (/root/reference/test_passkey.py:38,53-64): a 9994-token prompt, encoding mode,
    stride 96, budget 0.5, kv_policy roco -> budget'=5093, idx=5002, r_idx=4906 (README.md:205-214 prints
    ``50.05%(5002/9994)``), Hq=H=32, D=128: the dense 4906-token causal prefix, then 53 chunk steps of 96 queries per head over a
    cache oscillating 5002 <-> 5098 (easykv/easykv.py:367-503), then plain decode.  One layer (the per-layer work is
    independent); the layer-sharded run of the same driver is tests/test_hip_sharded_generate.py."""
    from easykv_amd import geometry
    assert geometry("encoding", 9994, 0.5, 96) == (5093, 5002, 4906)
    cfg = dict(budget=0.5, kv_policy="roco", max_new_tokens=3, temp_length=4, recent_ratio=0.1)
    res, tr, frac = _run_pair("encoding", 96, cfg, 1, 32, 32, 128, 9994, seed=9994)
    assert tr.report.strip() == "KV cache budget ratio: 50.05%(5002/9994)"
    assert tr.cache_len == 5002 + 3
    assert res == " ".join(str(t) for t in tr.result)


def test_config1_llama_prefill_stride8_full_geometry():
    """configs[1]: Llama2-7B prefill mode, S=4096, stride 8, budget 0.5, kv_policy roco -> budget'=2056, idx=2056, r_idx=2048,
    W=2064: the dense 2048-token causal prefix, then ALL 256 chunk steps of 8 queries per head over a cache
    oscillating 2056 <-> 2064 (easykv/easykv.py:367-503) through ``generate``, then plain decode.  One layer, Hq=H=32, D=128: a
    decoder stack issues one layer per call, 32 heads per launch, which the library runs as key-range splits of the MFMA chunk
    kernel + the scorer; the 32-layers-per-launch form of the same step (the logits-in-LDS kernel, what bench.py times) is
    compared with the oracle at this geometry by tests/test_hip_fullsize.py::test_chunk_steps_full_size_c2_shape."""
    from easykv_amd import geometry
    assert geometry("encoding", 4096, 0.5, 8) == (2056, 2056, 2048)
    cfg = dict(budget=0.5, kv_policy="roco", max_new_tokens=3, temp_length=4, recent_ratio=0.1)
    res, tr, frac = _run_pair("encoding", 8, cfg, 1, 32, 32, 128, 4096, seed=4096)
    assert tr.report.strip() == "KV cache budget ratio: 50.20%(2056/4096)"
    assert len([e for e in tr.evictions if e["kind"] == "per_head"]) == 255      # (the first chunk only fills the cache up to idx)
    assert tr.cache_len == 2056 + 3
    assert res == " ".join(str(t) for t in tr.result)


@pytest.mark.parametrize("policy,keep,hq,h,d,stride", [
    ("roco", True, 8, 4, 64, 40),        # GQA x2: 80 folded rows; the scored prefix walks its query blocks inside one launch pair
    ("h2o_head", False, 4, 4, 128, 96),  # 96 rows, head_dim 128
    ("roco", True, 8, 2, 128, 12),       # GQA x4: 48 folded rows (the 2 x 2 wave shape)
])
def test_streaming_wide_strides_through_generate(policy, keep, hq, h, d, stride):
    """Round 4: RoPE-on-read (streaming=True, easykv/llama_patch.py:310-327) at query blocks of 33..128 GQA-folded rows runs on the
    wide-block kernel's RoPE variants — dense prefix, keep_attention prefix (one launch pair per layer), chunk steps with the scorers of
    all layers deferred to one launch per forward — END TO END through ``easykv_amd.generate`` against the CPU oracle's ``generate``
    (encoding mode, then plain decode): printed line, every bound eviction decision, outputs."""
    cfg = dict(budget=0.5, kv_policy=policy, keep_attention=keep, streaming=True, max_new_tokens=3, temp_length=4, recent_ratio=0.1)
    res, tr, frac = _run_pair("encoding", stride, cfg, 2, hq, h, d, 700, seed=100 + stride, min_stable=0.9)
    assert res == " ".join(str(t) for t in tr.result)
