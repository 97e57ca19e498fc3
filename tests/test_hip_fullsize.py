"""Parity at BASELINE.json's full sizes (Llama2-7B shape: H=Hq=32, D=128, budget 2048 -> T=2049 decode; C2 chunk steps
S=4096, stride 8, budget 0.5 -> T=2064), a few layers, against the CPU oracle, plus size-independent properties.

A decision only binds the kernel when it is well defined: the oracle re-runs every selection under +-2e-5 relative
perturbations (the same probe oracle/gen_golden.py uses); a head stops being compared after its first unstable decision
(from there on two correct implementations may legitimately diverge).  The test requires >= 95 % of the decisions to be
stable AND every stable decision to match bit for bit."""
import numpy as np
import pytest
import torch

from tests.golden_util import out_close

pytestmark = pytest.mark.gpu


class Probe:
    """Stability probe of one selection: the oracle's own decision re-run on scores perturbed by +-2e-5 relative noise; a decision
    that moves under any of the draws is not well defined at fp32 and is excluded from the comparison.  TRIALS: a near-tie (margin
    far below the noise) escapes one draw with probability ~1/2, so three draws let 13 % of them through — and which way the
    oracle itself resolves such a tie can depend on the host CPU's libm / BLAS code path.  (One long-run test failed once in
    ~15 runs on the GPU pool and never again on other boxes or with poisoned device memory; an undetected near-tie is the one
    explanation that fits.)  Six draws let 1.6 % through."""
    PERT = 2e-5
    TRIALS = 6

    def __init__(self):
        self.gen = torch.Generator().manual_seed(7)
        self.last_unstable = None

    def __call__(self, fn, policy, s, q, c, args, ids):
        base = torch.sort(ids, dim=-1)[0]
        bad = torch.zeros(ids.shape[:-1], dtype=torch.bool)
        for _ in range(self.TRIALS):
            e1 = 1.0 + (torch.rand(s.shape, generator=self.gen) * 2 - 1) * self.PERT
            e2 = 1.0 + (torch.rand(s.shape, generator=self.gen) * 2 - 1) * self.PERT
            alt = fn(policy, s * e1, q * e2, c.clone(), *args)
            alt = alt.unsqueeze(-1) if alt.dim() < ids.dim() else alt
            bad |= (torch.sort(alt, dim=-1)[0] != base).any(dim=-1)
        self.last_unstable = bad


def _check_permutation(bank):
    m = bank.slot_of_pos.cpu().numpy()
    ref = np.arange(bank.cap)
    for l in range(m.shape[0]):
        for h in range(m.shape[1]):
            assert np.array_equal(np.sort(m[l, h]), ref)


@pytest.mark.parametrize("policy", ["roco", "h2o_head"])
def test_decode_full_size_bench_shape(policy):
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    L, H, D, budget, steps = 2, 32, 128, 2048, 20
    T = budget + 1
    g = torch.Generator().manual_seed(2024)
    k0, v0 = torch.randn(L, H, budget, D, generator=g).half(), torch.randn(L, H, budget, D, generator=g).half()
    warm = torch.rand(L, H, T, generator=g) * 1e-3        # synthetic warm state (bench.py starts the same way)
    banks = {"fused": KVBank(L, H, H, D, cap=T + 63), "split": KVBank(L, H, H, D, cap=T + 63)}
    for b in banks.values():
        b.load_rows(k0.cuda(), v0.cuda())
        b.state_init(T, 0)
        b.score_sum[:, :, :T] += warm.cuda()
        b.score_sq[:, :, :T] += (warm ** 2).cuda()
    states = []
    for l in range(L):
        st = O.LayerState(k=k0[l:l + 1].float(), v=v0[l:l + 1].float())
        st.s, st.q, st.c = O.init_state_decoding((H,), budget)
        st.s += warm[l]
        st.q += warm[l] ** 2
        states.append(st)
    alive = torch.ones(L, H, dtype=torch.bool)
    n_dec = n_stable = 0
    probe = Probe()
    O.SELECT_HOOK = probe
    try:
        for i in range(steps):
            q, k, v = (torch.randn(L, H, 1, D, generator=g).half() for _ in range(3))
            plan = StepPlan(policy=policy, phase="decode", evict=True, budget=budget)
            o_f, ids_f = banks["fused"].attend(StepPlan(policy=policy, phase="decode", evict=True, budget=budget, n_split=1), q.cuda(), k.cuda(), v.cuda())
            o_s, ids_s = banks["split"].attend(plan, q.cuda(), k.cuda(), v.cuda())
            assert torch.equal(ids_f, ids_s), "fused and split paths disagree"
            rw = int(budget * 0.3)
            for l in range(L):
                o_ref, ids_ref = O.layer_step(states[l], q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(),
                                              O.StepPlan(policy=policy, phase="decode", evict=True, budget=budget))
                unstable = probe.last_unstable
                assert out_close(o_f[l].float().cpu(), o_ref[0])
                got = ids_f[l, :, 0].cpu().long()
                same = got == ids_ref[:, 0]
                n_dec += int(alive[l].sum())
                n_stable += int((alive[l] & ~unstable).sum())
                assert bool(same[alive[l] & ~unstable].all()), (i, l)
                alive[l] &= ~unstable & same
                # window constraints hold for every head, stable or not
                if policy == "roco":
                    assert int(got.max()) < T - O.ROCO_TAIL
                else:
                    assert int(got.max()) < T - rw
    finally:
        O.SELECT_HOOK = None
    assert banks["fused"].step_plan(StepPlan(policy=policy, phase="decode", evict=True, budget=budget, n_split=1), 1)[1]
    assert n_stable >= 0.95 * n_dec, (n_stable, n_dec)
    for b in banks.values():
        assert b.n_slots == [budget] * L
        _check_permutation(b)


def test_chunk_steps_full_size_c2_shape():
    from easykv_amd import KVBank, StepPlan, geometry
    from oracle import easykv_oracle as O
    H, D, S, s = 32, 128, 4096, 8
    bp, idx, r_idx = geometry("encoding", S, 0.5, s)
    assert (bp, idx, r_idx) == (2056, 2056, 2048)
    recent, sink = int(bp * 0.1), 4
    g = torch.Generator().manual_seed(77)
    k0, v0 = torch.randn(1, H, r_idx, D, generator=g).half(), torch.randn(1, H, r_idx, D, generator=g).half()
    bank = KVBank(1, H, H, D, cap=idx + s)
    bank.load_rows(k0.cuda(), v0.cuda())
    bank.state_init(idx + s, 2, s)
    st = O.LayerState(k=k0.float(), v=v0.float())
    st.s, st.q, st.c = O.init_state_prefill((H,), idx, s, False)
    alive = torch.ones(H, dtype=torch.bool)
    probe = Probe()
    O.SELECT_HOOK = probe
    n_dec = n_stable = 0
    try:
        for step in range(6):
            q, k, v = (torch.randn(1, H, s, D, generator=g).half() for _ in range(3))
            t_now = bank.n_slots[0] + s
            kw = dict(policy="roco", phase="prefill", accumulate=t_now > idx, evict=t_now > idx, budget=bp, recent=recent, sink=sink, stride=s)
            out, ids = bank.attend(StepPlan(tova_head_mean=True, **kw), q.cuda(), k.cuda(), v.cuda())
            o_ref, ids_ref = O.layer_step(st, q.float(), k.float(), v.float(), O.StepPlan(tova_head_mean=True, **kw))
            assert out_close(out[0].float().cpu(), o_ref[0])
            if ids is not None:
                unstable = probe.last_unstable
                got = torch.sort(ids[0].cpu().long(), dim=-1)[0]
                ref = torch.sort(ids_ref, dim=-1)[0]
                same = (got == ref).all(dim=-1)
                n_dec += int(alive.sum())
                n_stable += int((alive & ~unstable).sum())
                assert bool(same[alive & ~unstable].all()), step
                alive &= ~unstable & same
                assert int(got.min()) >= sink and int(got.max()) < idx + s - O.ROCO_TAIL
    finally:
        O.SELECT_HOOK = None
    assert n_dec > 0 and n_stable >= 0.9 * n_dec, (n_stable, n_dec)
    assert bank.n_slots[0] == idx
    _check_permutation(bank)


def test_c4_shape_stride96_and_long_decode_rows():
    """Vicuna-16K passkey geometry (README.md:211): S=9994, stride 96, budget 0.5 -> idx 5002; two chunk steps at
    T = 5098 (96 query rows -> QPW=4 chunk kernel, generic scorer with 96 victims), then budgeted decode steps over a
    5003-wide row (ITEMS=24 decode variants), 2 heads only to keep the oracle fast."""
    from easykv_amd import KVBank, StepPlan, geometry
    from oracle import easykv_oracle as O
    H, D, S, s = 2, 128, 9994, 96
    bp, idx, r_idx = geometry("encoding", S, 0.5, s)
    assert idx == 5002
    recent, sink = int(bp * 0.1), 4
    g = torch.Generator().manual_seed(5)
    k0, v0 = torch.randn(1, H, idx, D, generator=g).half(), torch.randn(1, H, idx, D, generator=g).half()
    bank = KVBank(1, H, H, D, cap=idx + s)
    bank.load_rows(k0.cuda(), v0.cuda())
    bank.state_init(idx + s, 2, s)
    st = O.LayerState(k=k0.float(), v=v0.float())
    st.s, st.q, st.c = O.init_state_prefill((H,), idx, s, False)
    probe = Probe()
    O.SELECT_HOOK = probe
    try:
        for step in range(2):
            q, k, v = (torch.randn(1, H, s, D, generator=g).half() for _ in range(3))
            kw = dict(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=recent, sink=sink, stride=s)
            out, ids = bank.attend(StepPlan(**kw), q.cuda(), k.cuda(), v.cuda())
            o_ref, ids_ref = O.layer_step(st, q.float(), k.float(), v.float(), O.StepPlan(**kw))
            assert out_close(out[0].float().cpu(), o_ref[0])
            got, ref = torch.sort(ids[0].cpu().long(), dim=-1)[0], torch.sort(ids_ref, dim=-1)[0]
            ok = ~probe.last_unstable
            assert bool((got == ref).all(dim=-1)[ok].all())
            if not bool(ok.all()):
                pytest.skip("near-tie in this draw")
        # budgeted decode over the whole cache (auto-mode rules, easykv/easykv.py:670-748); the reference trims the
        # score rows to cache+1 columns first (:666-669)
        st.s, st.q, st.c = (x[..., :-(s - 1)].clone() for x in (st.s, st.q, st.c))
        for step in range(3):
            q, k, v = (torch.randn(1, H, 1, D, generator=g).half() for _ in range(3))
            kw = dict(policy="roco", phase="decode", accumulate=True, evict=True, budget=bp)
            out, ids = bank.attend(StepPlan(n_split=step % 2, **kw), q.cuda(), k.cuda(), v.cuda())
            o_ref, ids_ref = O.layer_step(st, q.float(), k.float(), v.float(), O.StepPlan(**kw))
            assert out_close(out[0].float().cpu(), o_ref[0])
            ok = ~probe.last_unstable
            assert bool((ids[0, :, 0].cpu().long() == ids_ref[:, 0])[ok].all())
            if not bool(ok.all()):
                pytest.skip("near-tie in this draw")
    finally:
        O.SELECT_HOOK = None
    assert bank.n_slots[0] == idx
    _check_permutation(bank)


@pytest.mark.parametrize("policy,idx,s", [("roco", 12400, 16), ("h2o_head", 12400, 16), ("roco", 16400, 48), ("roco", 30000, 8)])
def test_wide_score_rows_beyond_one_cus_lds(policy, idx, s):
    """A 16K-context cache at budget 0.75 (W = 12.4 k columns per head: the score rows + selection keys of one head are 198 KB,
    more than a CU's 160 KB of LDS — VERDICT r1 "size limits").  The generic scorer then keeps its working copies of S / Q / C in
    global scratch and only the keys in LDS; same arithmetic, so the usual bar holds: two scored chunk steps (stride 16, 16
    victims per head) and three budgeted decode steps against the oracle, eviction sets identical wherever the probe calls the
    decision well defined."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    H, D = 2, 64      # (idx 16400 / stride 48: a 32K context at budget 0.5 through the two-pass chunk kernels; 30000: near the limit)
    bp, recent, sink = idx, int(idx * 0.1), 4
    g = torch.Generator().manual_seed(31)
    k0, v0 = torch.randn(1, H, idx, D, generator=g).half(), torch.randn(1, H, idx, D, generator=g).half()
    bank = KVBank(1, H, H, D, cap=idx + s)
    bank.load_rows(k0.cuda(), v0.cuda())
    bank.state_init(idx + s, 2, s)
    st = O.LayerState(k=k0.float(), v=v0.float())
    st.s, st.q, st.c = O.init_state_prefill((H,), idx, s, False)
    probe = Probe()
    O.SELECT_HOOK = probe
    checked = 0
    try:
        for step in range(2):
            q, k, v = (torch.randn(1, H, s, D, generator=g).half() for _ in range(3))
            kw = dict(policy=policy, phase="prefill", accumulate=True, evict=True, budget=bp, recent=recent, sink=sink, stride=s)
            out, ids = bank.attend(StepPlan(**kw), q.cuda(), k.cuda(), v.cuda())
            o_ref, ids_ref = O.layer_step(st, q.float(), k.float(), v.float(), O.StepPlan(**kw))
            assert out_close(out[0].float().cpu(), o_ref[0])
            got, ref = torch.sort(ids[0].cpu().long(), dim=-1)[0], torch.sort(ids_ref, dim=-1)[0]
            ok = ~probe.last_unstable
            assert bool((got == ref).all(dim=-1)[ok].all()), step
            checked += int(ok.sum())
            if not bool(ok.all()):
                pytest.skip("near-tie in this draw")
        for name, row in (("score_sum", st.s), ("score_sq", st.q)):
            if policy == "roco" or name == "score_sum":
                assert torch.allclose(getattr(bank, name)[0, :, :idx].cpu(), row[..., :idx], rtol=2e-4, atol=1e-6), name
        st.s, st.q, st.c = (x[..., :-(s - 1)].clone() for x in (st.s, st.q, st.c))   # easykv/easykv.py:666-669
        for step in range(3):
            q, k, v = (torch.randn(1, H, 1, D, generator=g).half() for _ in range(3))
            kw = dict(policy=policy, phase="decode", accumulate=True, evict=True, budget=bp)
            out, ids = bank.attend(StepPlan(**kw), q.cuda(), k.cuda(), v.cuda())
            o_ref, ids_ref = O.layer_step(st, q.float(), k.float(), v.float(), O.StepPlan(**kw))
            assert out_close(out[0].float().cpu(), o_ref[0])
            ok = ~probe.last_unstable
            assert bool((ids[0, :, 0].cpu().long() == ids_ref[:, 0])[ok].all()), step
            checked += int(ok.sum())
            if not bool(ok.all()):
                pytest.skip("near-tie in this draw")
    finally:
        O.SELECT_HOOK = None
    assert checked == 5 * H
    assert bank.n_slots[0] == idx
    _check_permutation(bank)


def test_baseline_config0_geometry_decoding_budget200():
    """BASELINE.json configs[0]: decoding mode, budget=200, kv_policy='roco' (test_decoding.py:29-48 uses 300 / 150):
    W = 201, recent = int(200*0.3) = 60, k1 = 140; Llama2-7B head shape, prompt of 37 tokens that is never evicted."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    H, D, P, budget, steps = 32, 128, 37, 200, 260
    g = torch.Generator().manual_seed(200)
    qs, ks, vs = (torch.randn(1, H, P + steps, D, generator=g).half() for _ in range(3))
    bank = KVBank(1, H, H, D, cap=P + budget + 1)
    bank.load_rows(ks[:, :, :P].cuda(), vs[:, :, :P].cuda())
    bank.state_init(budget + 1, 0)
    st = O.LayerState(k=ks[:, :, :P].float(), v=vs[:, :, :P].float())
    st.s, st.q, st.c = O.init_state_decoding((H,), budget)
    alive = torch.ones(H, dtype=torch.bool)
    probe = Probe()
    O.SELECT_HOOK = probe
    n_dec = n_stable = 0
    try:
        for i in range(steps):
            t = P + i
            evict = (bank.n_slots[0] + 1 - P) > budget
            kw = dict(policy="roco", phase="decode", evict=evict, score_off=P, budget=budget)
            out, ids = bank.attend(StepPlan(n_split=i % 3, **kw), qs[:, :, t:t + 1].cuda().contiguous(), ks[:, :, t:t + 1].cuda().contiguous(),
                                   vs[:, :, t:t + 1].cuda().contiguous())
            o_ref, ids_ref = O.layer_step(st, qs[:, :, t:t + 1].float(), ks[:, :, t:t + 1].float(), vs[:, :, t:t + 1].float(), O.StepPlan(**kw))
            assert out_close(out[0].float().cpu(), o_ref[0])
            if evict:
                got = ids[0, :, 0].cpu().long() - P
                ok = ~probe.last_unstable
                same = got == ids_ref[:, 0]
                n_dec += int(alive.sum())
                n_stable += int((alive & ok).sum())
                assert bool(same[alive & ok].all()), i
                alive &= ok & same
                assert int(got.min()) >= 0 and int(got.max()) < budget + 1 - O.ROCO_TAIL     # prompt and last 10 never evicted
    finally:
        O.SELECT_HOOK = None
    assert bank.n_slots[0] == P + budget            # "KV cache budget ratio: ...(200/260)"
    assert n_stable >= 0.9 * n_dec, (n_stable, n_dec)
