"""API behaviours of the host mirror that the reference's users rely on (SURVEY.md §8 'quirks'), on the GPU."""
import contextlib
import io

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(length=120, L=2, Hq=4, H=4, D=32, seed=5):
    from oracle.fake_model import make_streams
    from tests.native_fake_model import NativeFakeModel
    return NativeFakeModel(*make_streams(L, Hq, H, D, length + 64, seed))


def _run(model, mode, stride, length, **cfg):
    import easykv_amd
    ids = torch.arange(length).view(1, -1) % 16
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res, cache = easykv_amd.generate(model, ids, dict(cfg, eos_token_ids=[-1]), kv_mode=mode, stride=stride, return_cache=True)
    return res, cache, buf.getvalue().strip()


def test_full_budget_takes_the_plain_prefill_branch():
    # easykv/easykv.py:372-377: float budget >= 1.0 or int budget >= length -> no eviction at all
    for budget in (1.0, 500):
        res, cache, line = _run(_model(), "encoding", 8, 120, budget=budget, kv_policy="roco", max_new_tokens=3)
        assert line == "KV cache budget ratio: 100.00%(120/120)"
        assert cache.get_seq_length() == 123


def test_auto_mode_asserts_like_the_reference():
    with pytest.raises(AssertionError):     # stride 1 (easykv/easykv.py:666-669)
        _run(_model(), "auto", 1, 120, budget=40, kv_policy="roco", max_new_tokens=2)
    with pytest.raises(AssertionError):     # h2o_head is not white-listed (:536-537)
        _run(_model(), "auto", 4, 120, budget=40, kv_policy="h2o_head", max_new_tokens=2)
    with pytest.raises(AssertionError):     # float budget (:222)
        _run(_model(), "auto", 4, 120, budget=0.5, kv_policy="roco", max_new_tokens=2)
    with pytest.raises(UnboundLocalError):  # random is broken in auto mode (:744)
        _run(_model(), "auto", 4, 120, budget=40, kv_policy="random", max_new_tokens=2, recent_ratio=0.3)


@pytest.mark.parametrize("mode,stride", [("decoding", 1), ("encoding", 4)])
def test_random_policy_keeps_the_budget(mode, stride):
    torch.manual_seed(0)
    length = 16 if mode == "decoding" else 100
    cfg = dict(budget=24 if mode == "decoding" else 0.5, kv_policy="random", max_new_tokens=40 if mode == "decoding" else 3)
    res, cache, line = _run(_model(length), mode, stride, length, **cfg)
    if mode == "decoding":
        assert line == "KV cache budget ratio: 60.00%(24/40)"
    else:
        assert line == "KV cache budget ratio: 52.00%(52/100)"


def test_ppl_int_budget_is_full_like_the_reference():
    # easykv/easykv.py:759: `if budget >= 1.0` catches every int budget (SURVEY.md probe 9)
    a, _, _ = _run(_model(64), "ppl", 4, 64, budget=40, kv_policy="roco")
    b, _, _ = _run(_model(64), "ppl", 4, 64, budget=1.0, kv_policy="roco")
    assert a == b


def test_layers_can_be_launched_one_by_one_or_batched():
    """The same step through per-layer launches (split path) and one batched launch (fused path) evicts the same slots."""
    from easykv_amd import KVBank, StepPlan
    L, Hq, H, D, T0, budget = 4, 8, 8, 128, 300, 299
    g = torch.Generator().manual_seed(11)
    k0, v0 = torch.randn(L, H, T0, D, generator=g).half().cuda(), torch.randn(L, H, T0, D, generator=g).half().cuda()
    banks = [KVBank(L, Hq, H, D, cap=T0 + 8) for _ in range(2)]
    for b in banks:
        b.load_rows(k0, v0)
        b.state_init(budget + 1, 0)
    for step in range(12):
        q, k, v = (torch.randn(L, h, 1, D, generator=g).half().cuda() for h in (Hq, H, H))
        plan = StepPlan(policy="roco", phase="decode", evict=True, budget=budget)
        plan_b = StepPlan(policy="roco", phase="decode", evict=True, budget=budget, n_split=1)   # one workgroup per head: fused
        o_b, ids_b = banks[0].attend(plan_b, q, k, v)
        outs, ids = [], []
        for l in range(L):
            o, i = banks[1].attend(plan, q[l:l + 1], k[l:l + 1], v[l:l + 1], layer_begin=l)
            outs.append(o)
            ids.append(i)
        assert torch.equal(ids_b, torch.cat(ids))
        assert torch.allclose(o_b.float(), torch.cat(outs).float(), atol=2e-3, rtol=1e-3)
    assert banks[0].step_plan(plan_b, 1)[1] is True        # batched, unsplit: one fused launch
    assert banks[1].step_plan(plan, 1, 0, 1)[1] is False   # single layer: split path


def test_overlapped_scorer_matches_inline_scorer():
    """Scorer on a side stream (attention + fold first, score/select/compaction asynchronously) == inline scorer."""
    from easykv_amd import KVBank, StepPlan
    L, Hq, H, D, T0, budget = 3, 8, 8, 128, 300, 299
    g = torch.Generator().manual_seed(12)
    k0, v0 = torch.randn(L, H, T0, D, generator=g).half().cuda(), torch.randn(L, H, T0, D, generator=g).half().cuda()
    banks = [KVBank(L, Hq, H, D, cap=T0 + 8) for _ in range(2)]
    for b in banks:
        b.load_rows(k0, v0)
        b.state_init(budget + 1, 0)
    for step in range(20):
        q, k, v = (torch.randn(L, h, 1, D, generator=g).half().cuda() for h in (Hq, H, H))
        plan = StepPlan(policy="roco", phase="decode", evict=True, budget=budget)
        for l in range(L):
            o0, i0 = banks[0].attend(plan, q[l:l + 1], k[l:l + 1], v[l:l + 1], layer_begin=l)
            o1, i1 = banks[1].attend(plan, q[l:l + 1], k[l:l + 1], v[l:l + 1], layer_begin=l, overlap_scorer=True)
            torch.cuda.synchronize()
            assert torch.equal(i0, i1) and torch.equal(o0, o1)
    banks[1].join()
    torch.cuda.synchronize()
    assert torch.equal(banks[0].slot_of_pos, banks[1].slot_of_pos)
    assert torch.equal(banks[0].score_sum, banks[1].score_sum)


def test_hipgraph_is_refused_on_a_layer_sharded_model():
    """A captured forward would contain the stage hand-off (dist.send / dist.recv): refused up front."""
    from easykv_amd.dist import LayerShard
    model = _model(16)
    model.layer_shard = LayerShard(0, 2, 2)
    with pytest.raises(ValueError, match="hipgraph"):
        _run(model, "decoding", 1, 16, budget=24, kv_policy="roco", max_new_tokens=4, hipgraph=True)
    model.layer_shard = LayerShard(0, 4, 2)          # more ranks than layers: some rank would own nothing
    with pytest.raises(ValueError, match="at least one"):
        _run(model, "decoding", 1, 16, budget=24, kv_policy="roco", max_new_tokens=4)


def test_a_forward_that_raises_mid_stack_does_not_wedge_the_bank():
    """layers that attended with defer=True before the model raised left 'pending' > 0 and every later deferred
    attend refused.  begin_forward / abort_step drop the half-open token step; the bank then behaves as if the aborted token
    had never been issued (the appended row sits in the slot the next token of that layer overwrites)."""
    from easykv_amd import KVBank, StepPlan, _lib
    L, Hq, H, D, T0, budget = 3, 8, 8, 64, 200, 199
    g = torch.Generator().manual_seed(21)
    k0, v0 = torch.randn(L, H, T0, D, generator=g).half().cuda(), torch.randn(L, H, T0, D, generator=g).half().cuda()
    banks = [KVBank(L, Hq, H, D, cap=T0 + 8) for _ in range(2)]
    for b in banks:
        b.load_rows(k0, v0)
        b.state_init(budget + 1, 0)
    plan = StepPlan(policy="roco", phase="decode", evict=True, budget=budget)
    junk = [torch.randn(1, h, 1, D, generator=g).half().cuda() for h in (Hq, H, H)]
    banks[1].attend(plan, *junk, layer_begin=0, defer=True)      # token step opened on layer 0 only ... and the model raises
    with pytest.raises(_lib.EkvError, match="1 of 3 layers"):
        banks[1].flush()
    assert banks[1].abort_step() == 1 and banks[1].abort_step() == 0
    for step in range(6):
        q, k, v = (torch.randn(L, h, 1, D, generator=g).half().cuda() for h in (Hq, H, H))
        ids = []
        for b in banks:
            for l in range(L):
                b.attend(plan, q[l:l + 1], k[l:l + 1], v[l:l + 1], layer_begin=l, defer=True)
            ids.append(b.flush().clone())
        assert torch.equal(ids[0], ids[1])
    torch.cuda.synchronize()
    assert torch.equal(banks[0].score_sum, banks[1].score_sum) and banks[0].n_slots == banks[1].n_slots
    ka, kb = banks[0].ordered_kv()[0], banks[1].ordered_kv()[0]
    assert torch.equal(ka, kb)
    banks[1].attend(plan, *junk, layer_begin=0, defer=True)
    banks[1].reset()                                              # reset() also forgets a half-open step
    assert banks[1]._defer is None
