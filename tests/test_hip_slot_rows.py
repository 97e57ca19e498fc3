"""Slot-indexed score rows (ABI 6, ``EKV_PHASE_SLOT_ROWS``): the one-launch decode step on rows indexed by physical slot — S / Q
rewritten, count base + birth written once per row, nothing moved on an eviction — against the same step on the ordered layout
(which the oracle tests pin: tests/test_hip_random_shapes.py, tests/test_hip_decode_parity.py).

Replaces the reference's re-packing of ``cache_attn_scores / cache_attn_scores_square / cache_counter`` after every eviction
(easykv/easykv.py:315-333) for the decode step; the decisions (:310-337: ``topk`` of std, ``argmin`` of mean — ties to the lower cache
index, i.e. the older entry) and the reported indices must be those of the ordered layout.

The two layouts form their softmax sums in a different order (thread-owned columns are physical rows in one, order indices in the
other), so scores differ in the last fp32 bit and a decision that hangs on that bit may fall either way.  Every disagreement is
therefore re-examined in fp64 from the ordered bank's state before the step: it must be a tie at fp32 resolution (relative gap
<= 1e-6 between the two picks, or a pick whose std sits on the boundary of roco's feasible set); the head is then dropped from the
comparison (its cache contents differ from there on)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(*shape, g):
    return torch.randn(*shape, generator=g).half()


def _heavy(k, g):
    """Keys with log-normal norms: a broad score distribution (a few columns take most of the attention), unlike i.i.d. normal keys —
    the selects' warm-start windows miss far more often on it and their exact fall-backs run."""
    return (k.float() * torch.exp(1.2 * torch.randn(*k.shape[:-1], 1, generator=g))).half()


def _banks(L, Hq, H, D, budget, seed, scatter=True, heavy=False):
    from easykv_amd import KVBank
    g = torch.Generator().manual_seed(seed)
    k0, v0 = _mk(L, H, budget, D, g=g), _mk(L, H, budget, D, g=g)
    if heavy:
        k0 = _heavy(k0, g)
    perm = torch.argsort(torch.rand(L, H, budget, generator=g), dim=-1).int()
    banks = []
    for slot in (False, True):
        b = KVBank(L, Hq, H, D, cap=budget + 9)
        b.use_slot_rows = slot
        b.load_rows(k0.cuda(), v0.cuda())
        if scatter:
            b.slot_of_pos[:, :, :budget] = perm.cuda()      # (rows were loaded in identity order: permute the map AND the rows alike)
            kk, vv = b.k.clone(), b.v.clone()
            idx = perm.cuda().long().unsqueeze(-1).expand(-1, -1, -1, D)
            b.k[:, :, :budget].scatter_(2, idx, kk[:, :, :budget])
            b.v[:, :, :budget].scatter_(2, idx, vv[:, :, :budget])
        b.state_init(budget + 1, 0)
        banks.append(b)
    return banks, g


def _near_tie(policy, S0, Q0, C0, keys, q, budget, va, vb, roco_tail=10):
    """fp64 re-evaluation of one head's decision from the ordered state before the step (easykv/easykv.py:287-337)."""
    D = keys.shape[-1]
    p = torch.softmax((q.double() @ keys.double().T) / D ** 0.5, -1).mean(0)
    S = S0.double() + p
    if policy != "roco":
        return abs(float(S[va] - S[vb])) <= 1e-6 * abs(float(S[va]))
    Q = Q0.double() + p * p
    c = C0.double() + 1.0
    mean = S / c
    sd = (Q / c - mean * mean).clamp_min(0).sqrt()
    sd[-roco_tail:] = 1e9
    k1 = budget - int(budget * 0.3)
    edge = torch.topk(sd, k1 + 1, largest=False).values[-2:]       # std at ranks k1 - 1 and k1: the boundary of the feasible set
    on_edge = any(abs(float(sd[v] - e)) <= 1e-6 * float(e) for v in (va, vb) for e in edge)
    return on_edge or abs(float(mean[va] - mean[vb])) <= 1e-6 * abs(float(mean[va]))


def _run(policy_schedule, L, Hq, H, D, budget, seed, min_alive=0.75, heavy=False):
    from easykv_amd import StepPlan
    (a, b), g = _banks(L, Hq, H, D, budget, seed, heavy=heavy)
    alive = torch.ones(L, H, dtype=torch.bool)
    T = budget + 1
    rep = Hq // H
    n_used = 0
    for policy, steps in policy_schedule:
        plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget, n_split=1)
        assert a.step_plan(plan, 1) == (1, True)                   # the one-launch decode step
        for i in range(steps):
            q, k, v = _mk(L, Hq, 1, D, g=g).cuda(), (_heavy(_mk(L, H, 1, D, g=g), g) if heavy else _mk(L, H, 1, D, g=g)).cuda(), _mk(L, H, 1, D, g=g).cuda()
            S0, Q0, C0 = a.score_sum.clone(), a.score_sq.clone(), a.score_cnt.clone()
            K0, _ = a.ordered_kv()
            oa, ia = a.attend(plan, q, k, v)
            ob, ib = b.attend(plan, q, k, v)
            n_used += int(any(b._slot_rows))
            same = (ia[:, :, 0] == ib[:, :, 0]).cpu()
            for l, h in (~same & alive).nonzero().tolist():
                keys = torch.cat([K0[l, h].float(), k[l, h].float()], 0)
                assert _near_tie(policy, S0[l, h, :T], Q0[l, h, :T], C0[l, h, :T], keys, q[l, h * rep:(h + 1) * rep, 0].float(), budget,
                                 int(ia[l, h, 0]), int(ib[l, h, 0])), (policy, i, l, h, int(ia[l, h, 0]), int(ib[l, h, 0]))
                alive[l, h] = False
            m = alive.repeat_interleave(rep, dim=1).cuda()
            assert torch.allclose(oa[m].float(), ob[m].float(), atol=1e-3, rtol=0), (policy, i)
    assert n_used == sum(s for _, s in policy_schedule), "the slot-indexed layout was not used"
    assert float(alive.float().mean()) >= min_alive, f"{int((~alive).sum())} of {alive.numel()} heads hit an fp32 tie"
    # back to the ordered layout: slot map, counts exactly; sums to fp32 rounding of a differently ordered softmax sum
    n = a.n_slots[0]
    assert a.n_slots == b.n_slots and a.extent == b.extent
    mk = alive.cuda()
    assert torch.equal(a.slot_of_pos[mk][:, :n], b.slot_of_pos[mk][:, :n])
    assert not any(b._slot_rows)                                   # (reading the state converted the layers back)
    assert torch.allclose(a.score_sum[mk], b.score_sum[mk], rtol=2e-5, atol=1e-9)
    if all(p == "roco" for p, _ in policy_schedule):               # (h2o_head / tova keep no Q / C rows: easykv.py:310-318 re-packs the sums only)
        assert torch.equal(a.score_cnt[mk], b.score_cnt[mk])
        assert torch.allclose(a.score_sq[mk], b.score_sq[mk], rtol=4e-5, atol=1e-12)
    for t in (b.slot_of_pos.cpu().numpy(),):                       # the map of EVERY head is still a permutation of the rows
        for l in range(L):
            for h in range(H):
                assert np.array_equal(np.sort(t[l, h]), np.arange(b.cap))
    return a, b


@pytest.mark.parametrize("policy,D,rep", [("roco", 128, 1), ("roco", 64, 2), ("h2o_head", 128, 4), ("tova", 128, 1)])
def test_long_run_on_the_slot_layout_equals_the_ordered_layout(policy, D, rep):
    """1200 evicting decode steps without ever leaving the slot-indexed layout (births run far past the cache length, every row is
    recycled many times), then the conversion back."""
    _run([(policy, 1200)], L=2, Hq=4 * rep, H=4, D=D, budget=120, seed=300 + D + rep)


def test_long_run_on_a_broad_score_distribution():
    """Keys with log-normal norms (a few columns take most of the probability mass): thresholds jump, the warm-started select misses
    and the full select / the bisection run — same victims as the ordered layout."""
    _run([("roco", 500)], L=2, Hq=4, H=4, D=128, budget=160, seed=41, heavy=True)
    _run([("roco", 60)], L=2, Hq=8, H=4, D=128, budget=2400, seed=42, heavy=True)


def test_extents_beyond_2304_rows():
    """The builds with 24 (4-wave) thread-owned columns: a budget of 3000."""
    _run([("roco", 60)], L=2, Hq=4, H=4, D=128, budget=3000, seed=21)
    _run([("h2o_head", 30)], L=1, Hq=8, H=4, D=128, budget=2500, seed=22)


def test_a_growing_protected_tail_takes_the_exact_bisection():
    """roco protects the 10 newest entries, h2o_head the newest 30 % of the budget: after a stretch of roco steps the newest 36
    entries are NOT consecutive births any more (roco evicted some of them), so the h2o_head steps cannot take ``nb - tail`` as the
    threshold — the counting check fails and the kernel finds the exact one.  (Not back to roco afterwards: the ordered h2o_head steps
    leave the Q / C rows unpacked, as the reference does, so the two banks' roco state would no longer describe the same entries.)"""
    _run([("roco", 150), ("h2o_head", 60)], L=2, Hq=4, H=4, D=128, budget=120, seed=11)


@pytest.mark.parametrize("budget_8w", [700, 2700])      # (2700: the 12-column build of the 8-wave kernel)
def test_llama_shape_launch_and_growth_steps(budget_8w):
    """The 8-wave kernel build (256..512 heads per launch; the tests above run the 4-wave one) at a budget of 700, starting BELOW the
    budget: the first steps append without evicting (rows come from the never-used part of the free list), then the steady state."""
    from easykv_amd import KVBank, StepPlan
    L, Hq, H, D, budget, fill = 16, 32, 32, 128, budget_8w, budget_8w - 10
    g = torch.Generator().manual_seed(5)
    k0, v0 = _mk(L, H, fill, D, g=g), _mk(L, H, fill, D, g=g)
    banks = []
    for slot in (False, True):
        b = KVBank(L, Hq, H, D, cap=budget + 9)
        b.use_slot_rows = slot
        b.load_rows(k0.cuda(), v0.cuda())
        b.state_init(fill + 1, 0)
        banks.append(b)
    a, b = banks
    diverged = torch.zeros(L, H, dtype=torch.bool)
    for i in range(40):
        evict = a.n_slots[0] >= budget
        plan = StepPlan(policy="roco", phase="decode", evict=evict, score_off=0, budget=budget)
        assert a.step_plan(plan, 1)[1]
        q, k, v = _mk(L, Hq, 1, D, g=g).cuda(), _mk(L, H, 1, D, g=g).cuda(), _mk(L, H, 1, D, g=g).cuda()
        oa, ia = a.attend(plan, q, k, v)
        ob, ib = b.attend(plan, q, k, v)
        assert all(b._slot_rows)
        if evict:
            diverged |= (ia[:, :, 0] != ib[:, :, 0]).cpu()
        assert torch.allclose(oa[~diverged.cuda()].float(), ob[~diverged.cuda()].float(), atol=1e-3, rtol=0), i
    assert int(diverged.sum()) <= 2          # (512 heads x 30 decisions: an fp32 tie is possible, more than a couple is a bug)
    keep = ~diverged.cuda()
    assert a.n_slots == b.n_slots == [budget] * L
    assert torch.equal(a.slot_of_pos[keep][:, :budget], b.slot_of_pos[keep][:, :budget])
    assert torch.equal(a.score_cnt[keep], b.score_cnt[keep])


def test_other_steps_convert_back_first():
    """A chunk step, a per-layer (deferred) decode step and a state read on slot-indexed layers all see the ordered layout."""
    from easykv_amd import StepPlan
    (a, b), g = _banks(2, 4, 4, 128, 120, seed=3)
    plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=120, n_split=1)
    for i in range(5):
        q, k, v = _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda()
        ia, ib = a.attend(plan, q, k, v)[1], b.attend(plan, q, k, v)[1]
        assert torch.equal(ia, ib)
    assert all(b._slot_rows)
    # a strided chunk step (q_len 4) on both
    cplan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=120, recent=12, sink=4, stride=4)
    q, k, v = _mk(2, 4, 4, 128, g=g).cuda(), _mk(2, 4, 4, 128, g=g).cuda(), _mk(2, 4, 4, 128, g=g).cuda()
    (oa, ia), (ob, ib) = a.attend(cplan, q, k, v), b.attend(cplan, q, k, v)
    assert not any(b._slot_rows)
    assert torch.equal(torch.sort(ia, -1).values, torch.sort(ib, -1).values) and torch.allclose(oa.float(), ob.float(), atol=1e-3, rtol=0)
    # decode again (slot layout), then one layer per call with the scorer deferred (ordered layout)
    q, k, v = _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda()
    assert torch.equal(a.attend(plan, q, k, v)[1], b.attend(plan, q, k, v)[1]) and all(b._slot_rows)
    dplan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=120)
    q, k, v = _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda()
    outs = []
    for bank in (a, b):
        o = torch.empty(2, 4, 1, 128, dtype=torch.float16, device="cuda")
        for l in range(2):
            bank.attend(dplan, q[l:l + 1], k[l:l + 1], v[l:l + 1], layer_begin=l, out=o[l:l + 1], defer=True)
        outs.append((o, bank.flush()))
    assert not any(b._slot_rows)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.allclose(outs[0][0].float(), outs[1][0].float(), atol=1e-3, rtol=0)
    assert torch.equal(a.slot_of_pos[:, :, :120], b.slot_of_pos[:, :, :120]) and torch.equal(a.score_cnt, b.score_cnt)


def test_a_caller_that_flips_the_layout_every_step_ends_up_on_the_ordered_one():
    """Reading the ordered state after every decode step costs two conversions per step: after four short stretches the bank stops
    entering the slot-indexed layout (results unchanged either way)."""
    from easykv_amd import StepPlan
    (a, b), g = _banks(2, 4, 4, 128, 120, seed=8)
    plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=120, n_split=1)
    used = []
    for i in range(10):
        q, k, v = _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda()
        ia, ib = a.attend(plan, q, k, v)[1], b.attend(plan, q, k, v)[1]
        used.append(all(b._slot_rows))
        assert torch.equal(ia, ib)
        assert torch.equal(a.score_cnt, b.score_cnt)          # (reads the ordered state: converts b back)
    assert used[:4] == [True] * 4 and used[-1] is False
    b.reset()
    assert b._slot_short == 0


def test_roco_at_a_budget_whose_feasible_set_reaches_into_the_protected_tail():
    """at budget 20 roco's feasible set (k1 = 14 of T = 21 entries) must reach into the 10 newest entries, which are
    protected by std = 1e9 sentinels only (easykv/easykv.py:318-321) — the arg-min over the mean may then evict one of them, after
    which the newest births are not consecutive and ``nb - tail`` is the wrong threshold.  The engine must not vouch for the tail
    (EKV_PHASE_SLOT_TAIL_OK) on such a step or after it; the kernel's counting check + bisection then gives the ordered layout's
    victims.  The sentinel entries tie exactly; both layouts break the tie to the older entry, so the ids must be EQUAL except at fp32
    near-ties of the mean."""
    from easykv_amd import StepPlan
    L, Hq, H, D, budget = 2, 4, 4, 128, 20
    (a, b), g = _banks(L, Hq, H, D, budget, seed=77)
    plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=budget, n_split=1)
    assert a.step_plan(plan, 1) == (1, True)
    T = budget + 1
    alive = torch.ones(L, H, dtype=torch.bool)
    evicted_from_tail = 0
    for i in range(300):
        q, k, v = _mk(L, Hq, 1, D, g=g).cuda(), _mk(L, H, 1, D, g=g).cuda(), _mk(L, H, 1, D, g=g).cuda()
        S0, C0 = a.score_sum.clone(), a.score_cnt.clone()
        K0, _ = a.ordered_kv()
        oa, ia = a.attend(plan, q, k, v)
        ob, ib = b.attend(plan, q, k, v)
        assert all(b._slot_rows)
        evicted_from_tail += int((ia[:, :, 0] >= T - 10).sum())
        same = (ia[:, :, 0] == ib[:, :, 0]).cpu()
        for l, h in (~same & alive).nonzero().tolist():
            keys = torch.cat([K0[l, h].float(), k[l, h].float()], 0)
            p = torch.softmax((q[l, h:h + 1, 0].double() @ keys.double().T) / D ** 0.5, -1).mean(0)
            mean = (S0[l, h, :T].double() + p) / (C0[l, h, :T].double() + 1.0)
            va, vb = int(ia[l, h, 0]), int(ib[l, h, 0])
            assert abs(float(mean[va] - mean[vb])) <= 1e-6 * abs(float(mean[va])), (i, l, h, va, vb)
            alive[l, h] = False
        m = alive.cuda()
        assert torch.allclose(oa[m].float(), ob[m].float(), atol=1e-3, rtol=0), i
    assert evicted_from_tail > 0, "no step evicted one of the 10 newest entries: the shape does not exercise the case"
    assert float(alive.float().mean()) >= 0.75
    mk = alive.cuda()
    assert torch.equal(a.slot_of_pos[mk][:, :budget], b.slot_of_pos[mk][:, :budget])
    assert torch.equal(a.score_cnt[mk], b.score_cnt[mk])


def test_the_thrash_guard_reopens_after_a_cool_down():
    """four short stretches early in a generate must not keep the bank on the ordered layout for the rest of the sequence."""
    from easykv_amd import StepPlan
    (a, b), g = _banks(2, 4, 4, 128, 120, seed=9)
    b.SLOT_COOL_DOWN = 12
    plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=0, budget=120, n_split=1)

    def step():
        q, k, v = _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda(), _mk(2, 4, 1, 128, g=g).cuda()
        assert torch.equal(a.attend(plan, q, k, v)[1], b.attend(plan, q, k, v)[1])
    for i in range(6):                       # a caller that reads the state after every step: the guard closes
        step()
        _ = b.score_cnt
    assert b._slot_short >= 4
    used = []
    for i in range(40):                      # ... then settles into pure decode
        step()
        used.append(all(b._slot_rows))
    assert not any(used[:8]) and all(used[-20:])
    assert torch.equal(a.score_cnt, b.score_cnt) and torch.equal(a.slot_of_pos[:, :, :120], b.slot_of_pos[:, :, :120])
