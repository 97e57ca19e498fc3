"""Boundary kernels of the C ABI: ordered gather (the HF legacy-tuple view), row import, and the reference-shaped
in-place compaction (easykv/easykv.py:56-82) — checked against the oracle's order-preserving delete."""
import numpy as np
import pytest
import torch

from tests.golden_util import out_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D", [32, 64, 128])
@pytest.mark.parametrize("k", [1, 7, 96])
def test_compact_inplace_equals_reference_delete(D, k):
    from easykv_amd import KVBank
    from oracle import easykv_oracle as O
    L, H, T = 2, 3, 700
    g = torch.Generator().manual_seed(D + k)
    kk = torch.randn(L, H, T, D, generator=g).half()
    vv = torch.randn(L, H, T, D, generator=g).half()
    ids = torch.stack([torch.stack([torch.randperm(T, generator=g)[:k].sort()[0] for _ in range(H)]) for _ in range(L)])
    bank = KVBank(L, H, H, D, cap=T)
    bank.load_rows(kk.cuda(), vv.cuda())
    bank.compact_inplace(ids.int().cuda())
    k_got, v_got = bank.ordered_kv()
    for l in range(L):
        k_ref = O.drop_kv_slots(kk[l:l + 1].float(), ids[l])
        v_ref = O.drop_kv_slots(vv[l:l + 1].float(), ids[l])
        assert torch.equal(k_got[l].float().cpu(), k_ref[0])
        assert torch.equal(v_got[l].float().cpu(), v_ref[0])
    assert bank.n_slots == [T - k] * L


def test_ordered_view_after_recycling_equals_reference_cache():
    """After many evictions rows are physically scrambled; the ordered view must equal the cache the reference would
    hold (birth order), bit for bit."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    L, Hq, H, D, P, budget, steps = 1, 4, 4, 64, 8, 40, 120
    g = torch.Generator().manual_seed(3)
    qs = torch.randn(L, Hq, P + steps, D, generator=g).half()
    ks = torch.randn(L, H, P + steps, D, generator=g).half()
    vs = torch.randn(L, H, P + steps, D, generator=g).half()
    bank = KVBank(L, Hq, H, D, cap=P + budget + 1)
    bank.load_rows(ks[:, :, :P].cuda(), vs[:, :, :P].cuda())
    bank.state_init(budget + 1, 0)
    st = O.LayerState(k=ks[:, :, :P].float(), v=vs[:, :, :P].float())
    st.s, st.q, st.c = O.init_state_decoding((H,), budget)
    for i in range(steps):
        t = P + i
        evict = (bank.n_slots[0] + 1 - P) > budget
        bank.attend(StepPlan(policy="h2o_head", phase="decode", evict=evict, score_off=P, budget=budget),
                    qs[:, :, t:t + 1].cuda().contiguous(), ks[:, :, t:t + 1].cuda().contiguous(), vs[:, :, t:t + 1].cuda().contiguous())
        O.layer_step(st, qs[:, :, t:t + 1].float(), ks[:, :, t:t + 1].float(), vs[:, :, t:t + 1].float(),
                     O.StepPlan(policy="h2o_head", phase="decode", evict=evict, score_off=P, budget=budget))
    k_got, v_got = bank.ordered_kv()
    assert torch.equal(k_got.float().cpu(), st.k) and torch.equal(v_got.float().cpu(), st.v)
    # the slot map is a permutation of the physical rows
    m = bank.slot_of_pos[0].cpu().numpy()
    for h in range(H):
        assert np.array_equal(np.sort(m[h]), np.arange(bank.cap))


def test_argument_errors_are_reported():
    from easykv_amd import KVBank, StepPlan
    bank = KVBank(1, 4, 4, 32, cap=64)
    q = torch.zeros(1, 4, 1, 32, dtype=torch.float16, device="cuda")
    bank.n_slots[0] = bank.cap          # no room for the new row
    with pytest.raises(ValueError):
        bank.attend(StepPlan(policy="full"), q, q, q)
    with pytest.raises(Exception):
        KVBank(1, 4, 4, 48, cap=64)     # unsupported head_dim


def test_long_unbudgeted_cache_decode_and_chunk():
    """kv_policy='full' over a 12k-slot cache (plain decode after an unbudgeted prefill, easykv/easykv.py:372-377, :508-526):
    no score rows are involved, any length must work."""
    import math
    from easykv_amd import KVBank, StepPlan
    L, Hq, H, D, T0 = 1, 8, 2, 128, 12000
    g = torch.Generator().manual_seed(9)
    k0, v0 = torch.randn(L, H, T0, D, generator=g).half().cuda(), torch.randn(L, H, T0, D, generator=g).half().cuda()
    bank = KVBank(L, Hq, H, D, cap=T0 + 40, scored=False)
    bank.load_rows(k0, v0)
    for n in (1, 5):
        q, k, v = (torch.randn(L, h, n, D, generator=g).half().cuda() for h in (Hq, H, H))
        out, ids = bank.attend(StepPlan(policy="full", phase="decode" if n == 1 else "prefill", accumulate=False), q, k, v)
        assert ids is None
        kk, vv = bank.ordered_kv()
        rep = Hq // H
        w = (q.float() @ kk.float().repeat_interleave(rep, 1).transpose(2, 3)) / math.sqrt(D)
        t = kk.shape[2]
        mask = torch.ones(n, t, dtype=torch.bool, device="cuda").tril(diagonal=t - n)
        w = w.masked_fill(~mask, float("-inf"))
        ref = torch.softmax(w, -1) @ vv.float().repeat_interleave(rep, 1)
        assert out_close(out.float(), ref), float((out.float() - ref).abs().max())
    assert bank.n_slots[0] == T0 + 6


@pytest.mark.parametrize("hq,h,n,t_prev,n_split", [(32, 8, 64, 0, 0), (32, 8, 64, 192, 0), (8, 8, 200, 100, 0), (8, 4, 96, 300, 3),
                                                  (16, 2, 40, 700, 2)])
@pytest.mark.parametrize("two_pass", [-1, 1])
def test_scored_step_with_several_query_blocks(hq, h, n, t_prev, n_split, two_pass):
    """rep x q_len > 128 folded rows -> several query blocks per head.  A block that ends before the chunk does stops
    at its own causal bound; the scorer must still see -inf (not stale workspace) on the positions after it — this is
    the keep_attention prefix of a GQA model (64 queries x rep 4)."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    d = 64
    g = torch.Generator().manual_seed(hq * 1000 + n)
    T = t_prev + n
    q = torch.randn(1, hq, n, d, generator=g).half()
    k = torch.randn(1, h, T, d, generator=g).half()
    v = torch.randn(1, h, T, d, generator=g).half()
    bank = KVBank(1, hq, h, d, cap=T + 64)
    if t_prev:
        bank.load_rows(k[:, :, :t_prev].cuda(), v[:, :, :t_prev].cuda())
    bank.state_init(T, 2, 1)
    # poison whatever workspace the allocator hands out next with finite values
    junk = torch.full((64 << 20,), 3.0, device="cuda")
    del junk
    out, _ = bank.attend(StepPlan(policy="roco", phase="prefill", accumulate=True, evict=False, n_split=n_split, two_pass=two_pass),
                         q.cuda(), k[:, :, t_prev:].cuda().contiguous(), v[:, :, t_prev:].cuda().contiguous())
    o_ref, p = O.attention_core(q.float(), k.float(), v.float(), O.causal_chunk_mask(n, T, torch.float32))
    pb = O.gqa_fold(p, h, hq // h)[0]
    assert out_close(out[0].float().cpu(), o_ref[0])
    assert torch.allclose(bank.score_sum[0, :, :T].cpu(), pb.sum(dim=-2), rtol=2e-5, atol=1e-7)
    assert torch.allclose(bank.score_sq[0, :, :T].cpu(), (pb ** 2).sum(dim=-2), rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("n,hq,h", [(1, 4, 4), (1, 8, 2), (8, 4, 4), (16, 8, 2)])
def test_phase_split_and_side_stream_scorer_equal_the_whole_step(n, hq, h):
    """ekv_step.phases: (attention + fold) then (scorer without fold) == the whole step, for decode and chunk steps; the
    engine's overlap_scorer mode (scorer on a side stream, next attention waits for it) gives the same trajectory."""
    from easykv_amd import KVBank, StepPlan
    d, t0, steps = 64, 300, 6
    g = torch.Generator().manual_seed(n * 100 + hq)
    k0, v0 = torch.randn(2, h, t0, d, generator=g).half().cuda(), torch.randn(2, h, t0, d, generator=g).half().cuda()
    qs = [torch.randn(2, hq, n, d, generator=g).half().cuda() for _ in range(steps)]
    ks = [torch.randn(2, h, n, d, generator=g).half().cuda() for _ in range(steps)]
    vs = [torch.randn(2, h, n, d, generator=g).half().cuda() for _ in range(steps)]
    if n == 1:
        kw = dict(policy="roco", phase="decode", evict=True, budget=t0, n_split=2)
    else:
        # two_pass=-1: every mode on the MFMA chunk kernel (whole small-row steps would otherwise take the logits-in-LDS kernel,
        # which is compared with this path in tests/test_hip_random_shapes.py — within tolerance, not bit for bit)
        kw = dict(policy="roco", phase="prefill", accumulate=True, evict=True, budget=t0 + n, recent=30, sink=4, stride=n, two_pass=-1)
    res = {}
    for mode in ("whole", "phases", "overlap"):
        bank = KVBank(2, hq, h, d, cap=t0 + n)
        bank.load_rows(k0, v0)
        if n == 1:
            bank.state_init(t0 + 1, 0)
        else:
            bank.state_init(t0 + n, 2, n)
        outs, idl = [], []
        for i in range(steps):
            if mode == "phases":
                out = torch.empty(2, hq, n, d, dtype=torch.float16, device="cuda")
                ids = torch.empty(2, h, n, dtype=torch.int32, device="cuda")
                bank.attend(StepPlan(**kw), qs[i], ks[i], vs[i], out=out, evict_ids=ids, phases=1 | 4)
                snap = out.clone()
                bank.attend(StepPlan(**kw), qs[i], ks[i], vs[i], out=out, evict_ids=ids, phases=8)
                assert torch.equal(snap, out)          # the scorer pass must not touch the folded output
            else:
                out, ids = bank.attend(StepPlan(**kw), qs[i], ks[i], vs[i], overlap_scorer=(mode == "overlap"))
            outs.append(out)
            idl.append(ids)
        bank.join()
        torch.cuda.synchronize()
        res[mode] = (torch.stack(outs).clone(), torch.stack(idl).clone(), bank.slot_of_pos.clone(), bank.score_sum.clone())
    for mode in ("phases", "overlap"):
        for a, b in zip(res["whole"], res[mode]):
            assert torch.equal(a, b), mode


@pytest.mark.parametrize("policy,hq,h,stream", [("roco", 8, 8, False), ("h2o_head", 8, 2, False), ("tova", 4, 4, True), ("recency", 4, 4, False)])
def test_deferred_scorer_per_layer_calls_equal_the_whole_step(policy, hq, h, stream):
    """ekv_step.defer_layers: one attention + fold call per layer (what a decoder stack issues) and ONE scorer launch for all
    layers at the end of the token == the whole step of every layer, bit for bit: outputs, evicted ids, score rows, slot maps."""
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import rope_tables
    L, d, t0, steps = 4, 64, 260, 7
    g = torch.Generator().manual_seed(hq * 10 + h)
    k0, v0 = torch.randn(L, h, t0, d, generator=g).half().cuda(), torch.randn(L, h, t0, d, generator=g).half().cuda()
    qs = [torch.randn(L, hq, 1, d, generator=g).half().cuda() for _ in range(steps)]
    ks = [torch.randn(L, h, 1, d, generator=g).half().cuda() for _ in range(steps)]
    vs = [torch.randn(L, h, 1, d, generator=g).half().cuda() for _ in range(steps)]
    res = {}
    for mode in ("whole", "deferred"):
        bank = KVBank(L, hq, h, d, cap=t0 + 1)
        if stream:
            bank.set_rope(*rope_tables(bank.cap + 8, d))
        bank.load_rows(k0, v0)
        bank.state_init(t0 + 1, 0)
        outs, idl = [], []
        for i in range(steps):
            plan = StepPlan(policy=policy, phase="decode", evict=True, budget=t0, n_split=3, streaming=stream,
                            accumulate=policy != "recency", range_start=5 if policy == "recency" else -1)
            if mode == "whole":
                out = torch.empty(L, hq, 1, d, dtype=torch.float16, device="cuda")
                ids = torch.full((L, h, 1), -1, dtype=torch.int32, device="cuda")
                for l in range(L):
                    o, ii = bank.attend(plan, qs[i][l:l + 1], ks[i][l:l + 1], vs[i][l:l + 1], layer_begin=l, out=out[l:l + 1], evict_ids=ids[l:l + 1])
            else:
                out = torch.empty(L, hq, 1, d, dtype=torch.float16, device="cuda")
                for l in range(L):
                    bank.attend(plan, qs[i][l:l + 1], ks[i][l:l + 1], vs[i][l:l + 1], layer_begin=l, out=out[l:l + 1], defer=True)
                    assert bank.n_slots[l] == t0          # nothing is evicted before the flush
                ids = bank.flush().clone()
            assert bank.n_slots == [t0] * L
            outs.append(out)
            idl.append(ids)
        torch.cuda.synchronize()
        res[mode] = (torch.stack(outs), torch.stack(idl), bank.slot_of_pos.clone(), bank.score_sum.clone(), list(bank.n_slots))
    for a, b in zip(res["whole"][:4], res["deferred"][:4]):
        assert torch.equal(a, b)
    assert res["whole"][4] == res["deferred"][4]


@pytest.mark.parametrize("policy,hq,h,d,stride,stream", [
    ("roco", 4, 4, 128, 96, False),       # wide kernel, two passes (column sums deferred)
    ("roco", 8, 2, 128, 16, False),       # GQA x4: 64 folded rows
    ("h2o_head", 4, 4, 64, 8, False),     # 16x16 kernel, one pass (logits + row statistics deferred)
    ("tova", 4, 4, 64, 8, False),         # tova: head-averaged last row
    ("roco", 4, 4, 128, 48, True),        # RoPE-on-read, wide kernel
    ("recency", 4, 4, 64, 8, False),      # range eviction
])
def test_deferred_scorer_chunk_steps_equal_the_whole_step(policy, hq, h, d, stride, stream):
    """ABI 5: ekv_step.defer_layers for CHUNK steps — per layer the attention launches + fold (the output the next layer waits for),
    ONE scorer launch over all layers at the end of the forward — equals the whole step of every layer bit for bit: outputs, evicted
    ids, score rows, slot maps.  (One layer per call is what a decoder stack issues; the per-layer scorer launch was more than half
    of a chunk step's time there.)"""
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import rope_tables
    L, t0, steps = 3, 1100, 4      # (T > 1024: one-layer calls of small strides stay off the logits-in-LDS kernel, whose arithmetic differs)
    g = torch.Generator().manual_seed(hq * 10 + h + stride)
    k0, v0 = torch.randn(L, h, t0, d, generator=g).half().cuda(), torch.randn(L, h, t0, d, generator=g).half().cuda()
    qs = [torch.randn(L, hq, stride, d, generator=g).half().cuda() for _ in range(steps)]
    ks = [torch.randn(L, h, stride, d, generator=g).half().cuda() for _ in range(steps)]
    vs = [torch.randn(L, h, stride, d, generator=g).half().cuda() for _ in range(steps)]
    res = {}
    for mode in ("whole", "deferred"):
        bank = KVBank(L, hq, h, d, cap=t0 + stride)
        if stream:
            bank.set_rope(*rope_tables(bank.cap + 8, d))
        bank.load_rows(k0, v0)
        bank.state_init(t0 + stride, 2, stride)
        outs, idl = [], []
        for i in range(steps):
            plan = StepPlan(policy=policy, phase="prefill", evict=True, accumulate=policy != "recency", budget=t0 + stride, recent=40, sink=4,
                            stride=stride, streaming=stream, tova_head_mean=policy == "tova", n_split=2,
                            range_start=4 if policy == "recency" else -1)
            out = torch.empty(L, hq, stride, d, dtype=torch.float16, device="cuda")
            if mode == "whole":
                ids = torch.full((L, h, stride), -1, dtype=torch.int32, device="cuda")
                for l in range(L):
                    bank.attend(plan, qs[i][l:l + 1], ks[i][l:l + 1], vs[i][l:l + 1], layer_begin=l, out=out[l:l + 1], evict_ids=ids[l:l + 1])
            else:
                for l in range(L):
                    bank.attend(plan, qs[i][l:l + 1], ks[i][l:l + 1], vs[i][l:l + 1], layer_begin=l, out=out[l:l + 1], defer=True)
                    assert bank.n_slots[l] == t0          # nothing is evicted before the flush
                ids = bank.flush().clone()
            assert bank.n_slots == [t0] * L
            outs.append(out)
            idl.append(torch.sort(ids, dim=-1)[0])
        torch.cuda.synchronize()
        res[mode] = (torch.stack(outs), torch.stack(idl), bank.slot_of_pos[:, :, :t0].clone(), bank.score_sum[:, :, :t0].clone(), list(bank.n_slots))
    for a, b in zip(res["whole"][:4], res["deferred"][:4]):
        assert torch.equal(a, b)
    assert res["whole"][4] == res["deferred"][4]
