"""Pins the CPU oracle to the reference: every golden vector was produced by the REAL reference
(oracle/gen_golden.py); the oracle must reproduce the eviction ids bit-identically, the attention
outputs to fp32 round-off, the cache lengths and the printed budget lines."""
import numpy as np
import pytest
import torch

from oracle import easykv_oracle as O
from oracle.fake_model import FakeAttnModel
from tests.golden_util import eos_ids, golden_names, load_golden, split_ids, split_outputs, trace_events


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_reference(name):
    g = load_golden(name)
    m = g["meta"]
    qs, ks, vs = g["streams"]
    eos, vocab = eos_ids(m)
    model = FakeAttnModel(qs, ks, vs, arch=m["arch"], streaming=m["streaming"], vocab=vocab)
    cfg = dict(m["config"], eos_token_ids=eos)
    ids = torch.arange(m["length"]).view(1, -1) % 16
    if m.get("rng_seed") is not None:      # kv_policy='random': the reference's draws come from the seeded global CPU generator
        torch.manual_seed(m["rng_seed"])
    tr = O.generate(model, ids, cfg, kv_mode=m["mode"], stride=m["stride"])
    kinds, ph, rg = trace_events(tr)
    assert np.array_equal(kinds, g["kinds"])
    ref_ph = split_ids(g)
    assert len(ph) == len(ref_ph)
    for a, b in zip(ph, ref_ph):
        assert np.array_equal(a, b)
    assert rg == [tuple(r) for r in g["ranges"].tolist()]
    assert tr.report == m["printed"]
    ref_out = split_outputs(g)
    assert len(model.outputs_log) == len(ref_out) == m["n_forwards"]
    for a, b in zip(model.outputs_log, ref_out):
        assert torch.allclose(a, b, rtol=0, atol=1e-6)
    if m["mode"] == "ppl":
        assert abs(tr.result - float(m["result"])) < 1e-9 * float(m["result"])


# Structural known answers printed in the reference's README: retained slots after
# strided prefill == idx of the geometry code (easykv/easykv.py:385-392, :773-780).
@pytest.mark.parametrize("length,stride,budget,kept,geom", [
    (9994, 96, 0.5, 5002, O.geometry_encoding),     # README.md:211, test_passkey.py:38
    (5144, 24, 0.5, 2576, O.geometry_encoding),     # README.md:153, test_passkey_NTK.py:44
    (10253, 96, 0.5, 5165, O.geometry_ppl),         # README.md:314, test_ppl.py:40
])
def test_readme_known_answers(length, stride, budget, kept, geom):
    _, idx, r_idx = geom(length, budget, stride)
    assert idx == kept
    assert (idx - r_idx) % stride == 0 and (length - idx) % stride == 0


def test_survey_geometry_table():
    # C2 / C3 / C5 geometry
    assert O.geometry_encoding(4096, 0.5, 8) == (2056, 2056, 2048)
    assert O.geometry_encoding(4096, 0.3, 16) == (1244, 1232, 1216)
    bp, idx, r = O.geometry_ppl(10253, 4096 / 10253, 96)
    assert (bp, idx, r) == (4192, 4109, 77)
    # auto mode fixes the cache at idx (probe 8): S=4096, budget 2048, s=64 -> 2112
    assert O.geometry_auto(4096, 2048, 64)[1] == 2112
    assert O.geometry_auto(4096, 2048, 8)[1] == 2056


def test_layer_step_equals_driver():
    """Per-layer fused step (what one HIP launch does) == all-layers-at-once driver order."""
    g = load_golden("dec_roco")
    m = g["meta"]
    qs, ks, vs = (x.float() for x in g["streams"])
    L, Hq, H, D = m["dims"]["L"], m["dims"]["Hq"], m["dims"]["H"], m["dims"]["D"]
    P, budget = m["length"], m["config"]["budget"]
    ref = split_ids(g)
    for l in range(L):
        pos = torch.arange(P)
        st = O.LayerState(k=ks[l][:, pos].unsqueeze(0), v=vs[l][:, pos].unsqueeze(0))
        st.s, st.q, st.c = O.init_state_decoding((H,), budget)
        e = 0
        for step in range(m["config"]["max_new_tokens"]):
            t = P + step
            gen = st.k.shape[2] + 1 - P
            plan = O.StepPlan(policy="roco", phase="decode", evict=gen > budget, score_off=P, budget=budget)
            _, ids = O.layer_step(st, qs[l][:, t:t + 1].unsqueeze(0), ks[l][:, t:t + 1].unsqueeze(0),
                                  vs[l][:, t:t + 1].unsqueeze(0), plan)
            if ids is not None:
                assert np.array_equal((ids + P).numpy().astype(np.int32), ref[e][l])
                e += 1
        assert e == len(ref)
