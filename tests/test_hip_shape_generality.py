"""Shapes the reference's attention takes and earlier rounds refused (VERDICT r5 "missing" #1): `repeat_kv` accepts ANY GQA factor
and any head_dim (/root/reference/easykv/llama_patch.py:19-29, :198-202).  Decode steps with factors 3 / 5 / 6 / 7 (the build of the
next power of two, padding heads out of the mean) and 12 / 16 (groups of 8 query heads on the split path), chunk steps with the same
factors, and head_dim 96 (12 live lanes of a 16-lane row group; the 16x16x32 MFMA chunk kernel with three k-steps) — every path
against the oracle on seeded inputs.  Decisions are asserted where the oracle's perturbation probe calls them well defined."""
import numpy as np
import pytest
import torch

from tests.golden_util import out_close
from tests.test_hip_fullsize import Probe

pytestmark = pytest.mark.gpu


def _mk(L, H, n, D, g):
    return torch.randn(L, H, n, D, generator=g).half()


DECODE_CASES = [
    # rep, D, H, L, budget, n_split, policy, stream
    (3, 128, 2, 2, 70, 0, "roco", False),
    (3, 64, 4, 2, 90, 2, "roco", True),
    (5, 128, 1, 3, 60, 0, "h2o_head", False),
    (6, 32, 2, 2, 120, 0, "roco", False),
    (7, 128, 4, 2, 80, 3, "roco", False),
    (7, 64, 2, 2, 80, 0, "tova", True),
    (12, 128, 2, 2, 70, 0, "roco", False),
    (16, 64, 1, 2, 50, 2, "h2o_head", False),
    (9, 32, 2, 1, 64, 0, "roco", True),
    (1, 96, 4, 2, 100, 0, "roco", False),
    (1, 96, 2, 2, 100, 3, "roco", True),
    (4, 96, 2, 2, 75, 0, "h2o_head", False),
    (3, 96, 2, 2, 75, 0, "roco", True),
    (8, 96, 1, 2, 60, 2, "tova", False),
    (2, 96, 3, 2, 200, 1, "roco", False),
]


@pytest.mark.parametrize("rep,D,H,L,budget,n_split,policy,stream", DECODE_CASES)
def test_decode_any_gqa_factor_and_head_dim_96(rep, D, H, L, budget, n_split, policy, stream):
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    Hq, P = H * rep, 5
    steps = budget + 25
    g = torch.Generator().manual_seed(rep * 1000 + D + budget)
    qs, ks, vs = _mk(L, Hq, P + steps, D, g), _mk(L, H, P + steps, D, g), _mk(L, H, P + steps, D, g)
    bank = KVBank(L, Hq, H, D, cap=P + budget + 1)
    cos = sin = None
    if stream:
        cos, sin = O.rope_tables(P + budget + 72, D)
        bank.set_rope(cos, sin)
    bank.load_rows(ks[:, :, :P].cuda(), vs[:, :, :P].cuda())
    bank.state_init(budget + 1, 0)
    sts = []
    for l in range(L):
        st = O.LayerState(k=ks[l:l + 1, :, :P].float(), v=vs[l:l + 1, :, :P].float())
        st.s, st.q, st.c = O.init_state_decoding((H,), budget)
        sts.append(st)
    alive = torch.ones(L, H, dtype=torch.bool)
    probe = Probe()
    O.SELECT_HOOK = probe
    n_checked = 0
    try:
        for i in range(steps):
            t = P + i
            evict = (bank.n_slots[0] + 1 - P) > budget
            kw = dict(policy=policy, phase="decode", evict=evict, score_off=P, budget=budget, streaming=stream)
            out, ids = bank.attend(StepPlan(n_split=n_split, **kw), qs[:, :, t:t + 1].cuda().contiguous(), ks[:, :, t:t + 1].cuda().contiguous(),
                                   vs[:, :, t:t + 1].cuda().contiguous())
            for l in range(L):
                if not bool(alive[l].all()):
                    continue
                o_ref, ids_ref = O.layer_step(sts[l], qs[l:l + 1, :, t:t + 1].float(), ks[l:l + 1, :, t:t + 1].float(), vs[l:l + 1, :, t:t + 1].float(),
                                              O.StepPlan(**kw), cos, sin)
                assert out_close(out[l].float().cpu(), o_ref[0]), (i, l, float((out[l].float().cpu() - o_ref[0]).abs().max()))
                if evict:
                    same = ids[l, :, 0].cpu().long() == ids_ref[:, 0] + P
                    ok = ~probe.last_unstable
                    assert bool(same[ok].all()), (i, l)
                    n_checked += int(ok.sum())
                    alive[l] &= ok & same
    finally:
        O.SELECT_HOOK = None
    assert n_checked >= 10, "too few well-defined decisions to be a meaningful test"
    m = bank.slot_of_pos.cpu().numpy()
    for l in range(L):
        for h in range(H):
            assert np.array_equal(np.sort(m[l, h]), np.arange(bank.cap))


CHUNK_CASES = [
    # rep, D, H, stride, idx, policy, stream, two_pass
    (3, 128, 2, 8, 200, "roco", False, 0),
    (3, 64, 2, 16, 300, "h2o_head", True, 0),
    (5, 128, 1, 4, 150, "roco", False, 0),
    (6, 64, 2, 24, 400, "roco", False, 0),       # 144 folded rows: two query blocks, exported logits
    (7, 32, 2, 3, 120, "tova", False, 0),
    (7, 128, 1, 16, 350, "roco", True, 0),
    (12, 64, 1, 8, 260, "roco", False, 0),
    (1, 96, 2, 8, 210, "roco", False, 0),
    (1, 96, 2, 64, 500, "roco", False, 1),       # statistics pass + exact pass of the 16x16 kernel, three k-steps
    (1, 96, 2, 64, 500, "roco", False, -1),
    (2, 96, 2, 48, 420, "h2o_head", False, 0),
    (4, 96, 1, 16, 330, "roco", True, 0),        # RoPE-on-read: partner pieces 6 lanes apart
    (1, 96, 3, 96, 640, "roco", True, 0),
    (3, 96, 2, 5, 170, "roco", False, 0),
    (1, 96, 1, 130, 700, "tova", False, 0),      # two query blocks of the 8-wave / 16-wave builds
]


class _ProbeAndCapture:
    """oracle.SELECT_HOOK: the stability probe of tests/test_hip_fullsize.py plus the rows the selection saw (tests/select_rule.py)."""

    def __init__(self):
        from tests.select_rule import Capture
        self.probe, self.cap = Probe(), Capture()

    def __call__(self, *args):
        self.probe(*args)
        self.cap(*args)


@pytest.mark.parametrize("rep,D,H,s,idx,policy,stream,two_pass", CHUNK_CASES)
def test_chunk_any_gqa_factor_and_head_dim_96(rep, D, H, s, idx, policy, stream, two_pass):
    """Evicting chunk steps of the encoding rules (easykv/easykv.py:443-499) from a cache of `idx` slots.  A decision is compared
    exactly where the probe calls it stable and roco's feasible set does not cut through a tied class (sentinels / NaN: torch takes
    an arbitrary subset there, tests/select_rule.py decides membership instead)."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    from tests.select_rule import feasible_classes, valid_victims
    Hq, L = H * rep, 2
    budget_p, recent, sink = idx + s // 2, int(idx * 0.2), 4
    g = torch.Generator().manual_seed(rep * 100 + D + s)
    k0, v0 = _mk(L, H, idx, D, g), _mk(L, H, idx, D, g)
    bank = KVBank(L, Hq, H, D, cap=idx + s)
    cos = sin = None
    if stream:
        cos, sin = O.rope_tables(idx + s + 72, D)
        bank.set_rope(cos, sin)
    bank.load_rows(k0.cuda(), v0.cuda())
    bank.state_init(idx + s, 2, s)
    sts = []
    for l in range(L):
        st = O.LayerState(k=k0[l:l + 1].float(), v=v0[l:l + 1].float())
        st.s, st.q, st.c = O.init_state_prefill((H,), idx, s, False)
        sts.append(st)
    hook = _ProbeAndCapture()
    O.SELECT_HOOK = hook
    alive = torch.ones(L, dtype=torch.bool)
    n_checked = 0
    k1 = max(budget_p - recent - sink, s)
    try:
        for step in range(4):
            q, k, v = _mk(L, Hq, s, D, g), _mk(L, H, s, D, g), _mk(L, H, s, D, g)
            kw = dict(policy=policy, phase="prefill", accumulate=True, evict=True, budget=budget_p, recent=recent, sink=sink, stride=s,
                      tova_head_mean=False, streaming=stream)
            out, ids = bank.attend(StepPlan(two_pass=two_pass, **kw), q.cuda(), k.cuda(), v.cuda())
            for l in range(L):
                if not bool(alive[l]):
                    continue
                o_ref, ids_ref = O.layer_step(sts[l], q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(), O.StepPlan(**kw), cos, sin)
                assert out_close(out[l].float().cpu(), o_ref[0]), (step, l, float((out[l].float().cpu() - o_ref[0]).abs().max()))
                got, ref = torch.sort(ids[l].cpu().long(), dim=-1)[0], torch.sort(ids_ref, dim=-1)[0]
                for h in range(H):
                    if bool(hook.probe.last_unstable[h]):
                        alive[l] = False
                        continue
                    tied = False
                    if policy == "roco":
                        c = hook.cap
                        forced, pool, need = feasible_classes(O.roco_std(c.s, c.q, c.c, sink)[h], k1)
                        tied = len(pool) != need
                        assert valid_victims(got[h].tolist(), (c.s / c.c)[h], forced, pool, need, s), (step, l, h)
                    if tied:
                        alive[l] = False      # (a valid outcome, but not necessarily torch's: the trajectories part here)
                    else:
                        assert bool((got[h] == ref[h]).all()), (step, l, h, got[h].tolist(), ref[h].tolist())
                        n_checked += 1
    finally:
        O.SELECT_HOOK = None
    assert n_checked >= 1
    assert bank.n_slots == [idx] * L
    m = bank.slot_of_pos.cpu().numpy()
    for l in range(L):
        for h in range(H):
            assert np.array_equal(np.sort(m[l, h]), np.arange(bank.cap))


@pytest.mark.parametrize("D,rep", [(96, 1), (96, 3), (128, 5)])
def test_boundary_copies_and_compaction_at_head_dim_96(D, rep):
    """ekv_scatter_rows / ekv_gather_ordered / ekv_compact_inplace with 12 pieces per row (21 rows per 256 threads)."""
    from easykv_amd import KVBank
    L, H, T = 2, 3, 301
    g = torch.Generator().manual_seed(D + rep)
    k0, v0 = _mk(L, H, T, D, g), _mk(L, H, T, D, g)
    bank = KVBank(L, H * rep, H, D, cap=T + 8)
    bank.load_rows(k0.cuda(), v0.cuda())
    ko, vo = bank.ordered_kv()
    assert torch.equal(ko.cpu(), k0) and torch.equal(vo.cpu(), v0)
    for kk in (1, 7):
        ids = torch.stack([torch.stack([torch.sort(torch.randperm(bank.n_slots[0], generator=g)[:kk])[0] for _ in range(H)]) for _ in range(L)]).int()
        t = bank.n_slots[0]
        bank.compact_inplace(ids.cuda())
        keep = torch.ones(L, H, t, dtype=torch.bool)
        keep.scatter_(2, ids.long(), False)
        k0 = k0[keep].view(L, H, t - kk, D)
        v0 = v0[keep].view(L, H, t - kk, D)
        assert torch.equal(bank.k[:, :, :t - kk].cpu(), k0) and torch.equal(bank.v[:, :, :t - kk].cpu(), v0)


def test_no_shape_the_reference_takes_is_refused():
    """ekv_step_check over GQA factors 1..16 x head_dim {32, 64, 96, 128} x decode / chunk steps."""
    from easykv_amd import KVBank, StepPlan
    import ctypes as C
    for D in (32, 64, 96, 128):
        for rep in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16):
            bank = KVBank(1, 2 * rep, 2, D, cap=2048)
            bank.n_slots = [1500]
            for n, plan in ((1, StepPlan(policy="roco", phase="decode", evict=True, budget=1500)),
                            (8, StepPlan(policy="roco", phase="prefill", evict=True, budget=1508, recent=150, sink=4, stride=8)),
                            (96, StepPlan(policy="h2o_head", phase="prefill", evict=True, budget=1596, recent=150, sink=4, stride=96))):
                st = bank.make_step(plan, n, 0, 1)
                assert bank.lib.ekv_step_check(C.byref(bank._bank), C.byref(st)) == 0, (D, rep, n)
