"""End-to-end: easykv_amd.generate (the host mirror of the reference's generate) over the HIP engine, against every
tie-free golden vector the real reference produced — all four modes, all policies, GQA, streaming, keep_attention.

Checks: the reference's printed budget line, the evicted position sets of every forward (bit-identical), every
forward's attention outputs (<= 1e-3), and the perplexity."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests.golden_util import golden_names, load_golden, split_ids, split_outputs

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-3   # north_star: within 1e-3 on fp16 outputs (outputs are fp16: half-ulp at |o| in [1,2) is 4.9e-4)


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
    model = NativeFakeModel(*g["streams"], arch=m["arch"])
    cfg = dict(m["config"], eos_token_ids=[-1], _record_evictions=True)
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
        assert torch.allclose(a, b, rtol=OUT_TOL / 2, atol=OUT_TOL), (f, float((a - b).abs().max()))
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
