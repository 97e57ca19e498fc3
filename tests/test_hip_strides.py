"""ABI 8: q / k_new / v_new / out addressed by (token stride, head stride) — SURVEY.md §8b "raw device pointers + explicit strides".
The reference's patched attention views the projections' [1, n, H*D] output as [1, H, n, D] (llama_patch.py:176-182) and transposes
the result back for o_proj (:230-232); until round 5 every chunk forward copied q, k and v dense in front of the step.  A step fed
with those VIEWS (and writing `out` token-major) must be bit-identical — outputs, evicted ids, slot map, score rows, K/V rows — to
the same step on dense copies, on every chunk kernel."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # name: (D, rep, H, stride, T0, two_pass, streaming, n_split, policy)
    "chunk_lds": (128, 1, 4, 8, 600, 0, False, 0, "roco"),
    "chunk16_tail": (64, 2, 2, 16, 500, -1, False, 1, "roco"),
    "chunk16_split": (64, 2, 2, 16, 500, -1, False, 2, "h2o_head"),
    "chunk16_two_pass": (32, 2, 2, 24, 400, 1, False, 2, "roco"),
    "chunk16_rope": (32, 1, 2, 16, 300, 0, True, 0, "roco"),
    "wide_two_pass_tail": (128, 1, 4, 64, 700, 1, False, 1, "roco"),
    "wide_two_pass_split": (128, 4, 2, 24, 900, 1, False, 2, "h2o_head"),
    "resident": (128, 4, 2, 16, 900, 0, False, 0, "roco"),           # the logits-resident one-launch step (64 folded rows, T <= 1280)
    "wide_rope": (128, 1, 2, 96, 800, 0, True, 0, "roco"),
    "wide_unscored": (64, 1, 4, 128, 0, 0, False, 0, "full"),        # the dense prefix: every row is one of the launch's own
    "d96": (96, 1, 2, 16, 300, 0, False, 0, "roco"),
}


def _views(L, heads, n, D, g):
    """dense [L, heads, n, D] fp16 and an HF-style view of the same values: [L, n, heads * D] storage seen as [L, heads, n, D]."""
    base = torch.randn(L, n, heads * D, generator=g).half().cuda()
    view = base.view(L, n, heads, D).transpose(1, 2)
    return view.contiguous(), view


@pytest.mark.parametrize("defer", [False, True], ids=["immediate", "deferred"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_strided_rows_equal_dense_copies(name, defer):
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import rope_tables
    D, rep, H, s, T0, two_pass, stream, n_split, policy = CASES[name]
    if defer and (policy == "full" or name == "chunk_lds"):
        pytest.skip("the deferred scorer is for scored steps that are not one launch already")
    Hq, L = H * rep, (3 if defer else 1)
    g = torch.Generator().manual_seed(len(name) * 7 + s)
    banks = []
    for _ in range(2):
        b = KVBank(L, Hq, H, D, cap=T0 + 2 * s + 8)
        if stream:
            b.set_rope(*rope_tables(T0 + 2 * s + 72, D))
        if T0:
            gk = torch.Generator().manual_seed(99)
            b.load_rows(torch.randn(L, H, T0, D, generator=gk).half().cuda(), torch.randn(L, H, T0, D, generator=gk).half().cuda())
        if policy != "full":
            b.state_init(T0 + s, 2, s)
        banks.append(b)
    for step in range(2):
        evict = policy != "full"
        plan_kw = dict(policy=policy, phase="prefill", accumulate=policy != "full", evict=evict, budget=T0 + s // 2, recent=int(T0 * 0.2), sink=4,
                       stride=s, streaming=stream, n_split=n_split, two_pass=two_pass)
        qd, qv = _views(L, Hq, s, D, g)
        kd, kv = _views(L, H, s, D, g)
        vd, vv = _views(L, H, s, D, g)
        assert not qv.is_contiguous() and qv.stride(2) == Hq * D and qv.stride(1) == D
        outs, ids = [], []
        for b, (q, k, v), strided in ((banks[0], (qd, kd, vd), False), (banks[1], (qv, kv, vv), True)):
            plan = StepPlan(**plan_kw)
            if defer:
                o = torch.empty(L, s, Hq, D, dtype=torch.float16, device="cuda").transpose(1, 2) if strided else torch.empty(L, Hq, s, D, dtype=torch.float16, device="cuda")
                for l in range(L):
                    b.attend(plan, q[l:l + 1], k[l:l + 1], v[l:l + 1], layer_begin=l, defer=True, out=o[l:l + 1])
                e = b.flush()
            else:
                o = torch.empty(L, s, Hq, D, dtype=torch.float16, device="cuda").transpose(1, 2) if strided else None
                o, e = b.attend(plan, q, k, v, out=o)
                if strided:
                    assert o.stride(2) == Hq * D and o.transpose(1, 2).is_contiguous()
            outs.append(o)
            ids.append(e)
        assert torch.equal(outs[0], outs[1]), (name, step, float((outs[0].float() - outs[1].float()).abs().max()))
        if evict:
            assert torch.equal(ids[0], ids[1])
    a, b = banks
    assert a.n_slots == b.n_slots
    t = a.n_slots[0]
    assert torch.equal(a.slot_of_pos, b.slot_of_pos)
    if policy != "full":
        assert torch.equal(a.score_sum[:, :, :t], b.score_sum[:, :, :t]) and torch.equal(a.score_sq[:, :, :t], b.score_sq[:, :, :t])
    ka, va = a.ordered_kv()
    kb, vb = b.ordered_kv()
    assert torch.equal(ka, kb) and torch.equal(va, vb)


def test_stride_arguments_are_validated():
    from easykv_amd import KVBank, StepPlan
    bank = KVBank(1, 4, 4, 64, cap=256)
    bank.n_slots = [100]
    st = bank.make_step(StepPlan(policy="full", phase="prefill", accumulate=False), 8, 0, 1)
    check = lambda: bank.lib.ekv_step_check(C.byref(bank._bank), C.byref(st))
    assert check() == 0
    st.q_token_stride, st.q_head_stride = 4 * 64, 64
    assert check() == 0
    st.q_token_stride = 4 * 64 + 4          # not a multiple of 8 halfs
    assert check() == -1
    st.q_token_stride, st.q_head_stride = 32, 64        # rows would overlap
    assert check() == -1
    st.q_token_stride, st.q_head_stride = 0, 0
    st.out_token_stride = -8
    assert check() == -1
    # q_len = 1: only head rows head_dim apart (any token stride)
    st1 = bank.make_step(StepPlan(policy="full", phase="decode", accumulate=False), 1, 0, 1)
    st1.q_token_stride, st1.q_head_stride = 4 * 64, 64
    assert bank.lib.ekv_step_check(C.byref(bank._bank), C.byref(st1)) == 0
    st1.q_head_stride = 128
    assert bank.lib.ekv_step_check(C.byref(bank._bank), C.byref(st1)) == -2
