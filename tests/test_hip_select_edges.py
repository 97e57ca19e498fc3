"""Edge cases of the reference's roco selection that tie-free fixtures never reach.

(a) NaN standard deviations.  `std = sqrt(Q/C - (S/C)**2)` (/root/reference/easykv/easykv.py:320, :471) is NaN whenever rounding
    leaves the radicand negative; `topk(largest=False)` ranks NaN last, behind the 1e9 sentinels the
    reference writes over the newest 10 (and the first `sink`) entries (:321, :472-473).  The score rows are seeded with columns
    whose radicand is clearly negative — and whose MEAN is the lowest of the row, so a kernel that let one into the feasible set
    would evict it — and the feasible-set size k1 is placed (i) inside the real keys, (ii) exactly at real + all sentinels,
    (iii) inside the sentinels, (iv) inside the NaNs.
(b) Ties.  Where k1 cuts through a class of equal keys (sentinels, NaNs) torch's topk takes an ARBITRARY subset (appendix A, probe
    T), so the reference's decision is a set of valid outcomes: the victims must be the arg-min-mean of the forced members plus SOME
    admissible choice from the tied class.  `valid_victims` decides membership exactly.  The two `*_ties` goldens of the real
    reference (k1 reaching into the sentinels at a small budget) run on the GPU under that rule: printed line, outputs and victim
    membership at every step.
Every kernel that selects is driven: the one-launch decode step on both score-row layouts, the split decode path, the
logits-in-LDS chunk kernel, the generic scorer, the scorer as the tail of the 16x16 chunk kernel and of the wide column-sum pass."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests.golden_util import golden_names, load_golden, split_outputs, out_close
from tests.select_rule import Capture, feasible_classes, seed_rows, valid_victims

pytestmark = pytest.mark.gpu


def _mk(L, H, n, D, g):
    return torch.randn(L, H, n, D, generator=g).half()


# ---------------------------------------------------------------------------------------------------------------------
# (a) decode
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reach,n_nan", [("real", 10), ("all_sentinels", 91), ("some_sentinels", 86), ("some_nans", 94)])
@pytest.mark.parametrize("path", ["fused_ordered", "fused_slot_rows", "split"])
def test_decode_nan_std_and_sentinels(path, reach, n_nan):
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    L, H, D, budget = 2, 4, 64, 300
    W, rw = budget + 1, int(budget * 0.3)
    k1 = budget - rw
    n_real = W - 10 - n_nan
    assert {"real": k1 <= n_real, "all_sentinels": k1 == n_real + 10, "some_sentinels": n_real < k1 < n_real + 10,
            "some_nans": k1 > n_real + 10}[reach]
    g = torch.Generator().manual_seed(n_nan * 3 + len(path))
    k0, v0 = _mk(L, H, budget, D, g), _mk(L, H, budget, D, g)
    bank = KVBank(L, H, H, D, cap=W + 7)
    bank.use_slot_rows = path == "fused_slot_rows"
    bank.load_rows(k0.cuda(), v0.cuda())
    bank.state_init(W, 0)
    rows = [seed_rows(H, W, budget, n_nan, g, nan_hi=W - 10) for _ in range(L)]      # (NaN columns outside the newest-10 window)
    for l in range(L):
        bank.score_sum[l, :, :W] = rows[l][0].cuda()
        bank.score_sq[l, :, :W] = rows[l][1].cuda()
        bank.score_cnt[l, :, :W] = rows[l][2].cuda()
    n_split = 2 if path == "split" else 1      # (unsplit heads: the whole step is one launch)
    kw = dict(policy="roco", phase="decode", evict=True, score_off=0, budget=budget)
    assert bank.step_plan(StepPlan(n_split=n_split, **kw), 1)[1] == (path != "split")
    cap = Capture()
    O.SELECT_HOOK = cap
    try:
        for step in range(3):
            q, k, v = _mk(L, H, 1, D, g), _mk(L, H, 1, D, g), _mk(L, H, 1, D, g)
            kord, vord = (t.float().cpu() for t in bank.ordered_kv())
            state = [t[:, :, :W].cpu().clone() for t in (bank.score_sum, bank.score_sq, bank.score_cnt)]
            if path == "fused_slot_rows":
                bank.use_slot_rows = True
            out, ids = bank.attend(StepPlan(n_split=n_split, **kw), q.cuda(), k.cuda(), v.cuda())
            if path == "fused_slot_rows":
                assert all(bank._slot_rows), "the step was meant to run on the slot-indexed layout"
            for l in range(L):
                st = O.LayerState(k=kord[l:l + 1], v=vord[l:l + 1], s=state[0][l], q=state[1][l], c=state[2][l])
                o_ref, ids_ref = O.layer_step(st, q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(), O.StepPlan(**kw))
                assert out_close(out[l].float().cpu(), o_ref[0])
                std = O.roco_std(cap.s, cap.q, cap.c)
                mean = cap.s / cap.c
                for h in range(H):
                    assert int(torch.isnan(std[h]).sum()) >= n_nan - 3, "the seeded columns must still be NaN at selection time"
                    forced, pool, need = feasible_classes(std[h], k1)
                    got = int(ids[l, h, 0])
                    assert valid_victims([got], mean[h], forced, pool, need, 1), (step, l, h, got, reach)
                    if len(pool) == need:      # the reference's decision is unique: the oracle's must be ours
                        assert got == int(ids_ref[h, 0]), (step, l, h, reach)
    finally:
        O.SELECT_HOOK = None


# ---------------------------------------------------------------------------------------------------------------------
# (a) prefill: k = stride victims
# ---------------------------------------------------------------------------------------------------------------------
PREFILL_PATHS = {
    # name: (D, rep, stride, idx, two_pass, n_split, layers, expected ekv_step_info subset)
    "chunk_lds": (128, 1, 8, 600, 0, 0, 2, dict(fused=1)),                     # logits in LDS, whole step one launch
    "generic_scorer": (64, 2, 16, 500, -1, 2, 2, dict(fused=0, two_pass=0)),   # exported logits, split heads, stand-alone scorer
    "chunk_tail": (64, 1, 16, 500, -1, 1, 2, dict(fused=1, wide=0)),           # scorer as the tail of the 16x16 chunk kernel
    "wide_tail": (128, 1, 64, 700, 1, 1, 2, dict(two_pass=1, wide=1)),         # scorer as the tail of the wide column-sum pass
    "two_pass_16x16": (32, 2, 24, 400, 1, 2, 2, dict(two_pass=1, wide=0)),     # statistics + exact pass, stand-alone scorer
}


@pytest.mark.parametrize("reach", ["real", "all_sentinels", "some_sentinels", "some_nans"])
@pytest.mark.parametrize("path", sorted(PREFILL_PATHS))
def test_prefill_nan_std_and_sentinels(path, reach):
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    D, rep, s, idx, two_pass, n_split, L, expect = PREFILL_PATHS[path]
    H = 4
    Hq, W, sink = H * rep, idx + s, 4
    n_sent = sink + 10
    n_nan = 60
    n_real = W - n_sent - n_nan + 0      # (the NaN columns are drawn from [sink, idx): none of them under a sentinel)
    k1 = {"real": n_real - 20, "all_sentinels": n_real + n_sent, "some_sentinels": n_real + 6, "some_nans": n_real + n_sent + 7}[reach]
    budget_p = idx + s // 2
    recent = budget_p - sink - k1
    assert recent >= 0 and max(budget_p - recent - sink, s) == k1
    g = torch.Generator().manual_seed(len(path) * 17 + len(reach))
    k0, v0 = _mk(L, H, idx, D, g), _mk(L, H, idx, D, g)
    bank = KVBank(L, Hq, H, D, cap=W)
    bank.load_rows(k0.cuda(), v0.cuda())
    bank.state_init(W, 2, s)
    n_check = min(L, 2)
    rows = []
    for l in range(L):
        # NaN columns outside the sink window and the newest-10 window (a sentinel overrides a NaN, easykv.py:472-473)
        # (counts of at least 6 strides: the step adds `stride` to every count, and Q/C - (S/C)^2 of the seeded columns stays
        #  negative only while C / (C + stride) > 0.8)
        sr, qr, cr, _ = seed_rows(H, W, idx, n_nan, g, c_lo=6 * s, c_hi=6 * s + 50, nan_lo=sink, nan_hi=min(idx, W - 10))
        cr[:, idx:] = -torch.arange(s, dtype=torch.float32)      # the count tail of easykv.py:416
        rows.append((sr, qr, cr))
        bank.score_sum[l, :, :W] = sr.cuda()
        bank.score_sq[l, :, :W] = qr.cuda()
        bank.score_cnt[l, :, :W] = cr.cuda()
    kw = dict(policy="roco", phase="prefill", accumulate=True, evict=True, budget=budget_p, recent=recent, sink=sink, stride=s)
    info = bank.step_info(StepPlan(n_split=n_split, two_pass=two_pass, **kw), s)
    for key, val in expect.items():
        assert info[key] == val, (path, info)
    q, k, v = _mk(L, Hq, s, D, g), _mk(L, H, s, D, g), _mk(L, H, s, D, g)
    out, ids = bank.attend(StepPlan(n_split=n_split, two_pass=two_pass, **kw), q.cuda(), k.cuda(), v.cuda())
    cap = Capture()
    O.SELECT_HOOK = cap
    try:
        for l in range(n_check):
            st = O.LayerState(k=k0[l:l + 1].float(), v=v0[l:l + 1].float(), s=rows[l][0].clone(), q=rows[l][1].clone(), c=rows[l][2].clone())
            o_ref, ids_ref = O.layer_step(st, q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(), O.StepPlan(**kw))
            assert out_close(out[l].float().cpu(), o_ref[0])
            std = O.roco_std(cap.s, cap.q, cap.c, sink)
            mean = cap.s / cap.c
            for h in range(H):
                assert int(torch.isnan(std[h]).sum()) == n_nan
                forced, pool, need = feasible_classes(std[h], k1)
                got = ids[l, h].cpu().tolist()
                assert valid_victims(got, mean[h], forced, pool, need, s), (path, reach, l, h)
                if len(pool) == need:
                    assert sorted(got) == sorted(ids_ref[h].tolist()), (path, reach, l, h)
    finally:
        O.SELECT_HOOK = None
    m = bank.slot_of_pos.cpu().numpy()
    for l in range(n_check):
        for h in range(H):
            assert np.array_equal(np.sort(m[l, h]), np.arange(bank.cap))


# ---------------------------------------------------------------------------------------------------------------------
# (b) the reference's own tie fixtures
# ---------------------------------------------------------------------------------------------------------------------
def _tie_cases():
    return [n for n in golden_names() if not load_golden(n)["meta"]["tie_free"]]


@pytest.mark.parametrize("name", _tie_cases())
def test_tie_goldens_printed_line_and_outputs_until_the_first_tie(name):
    """End to end through easykv_amd.generate: the printed budget line (cache sizes do not depend on which tied entry goes), the
    returned text, and every forward's attention outputs up to and including the first evicting forward (whose inputs no eviction
    has touched yet)."""
    import easykv_amd
    from tests.native_fake_model import NativeFakeModel
    g = load_golden(name)
    m = g["meta"]
    model = NativeFakeModel(*g["streams"], arch=m["arch"], vocab=m.get("vocab", 16))
    cfg = dict(m["config"], eos_token_ids=m.get("eos_token_ids", [-1]), _record_evictions=True)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res, cache = easykv_amd.generate(model, torch.arange(m["length"]).view(1, -1) % 16, cfg, kv_mode=m["mode"], stride=m["stride"], return_cache=True)
    assert buf.getvalue().strip() == m["printed"]
    assert res == m["result"]
    ref_out = split_outputs(g)
    assert len(model.outputs_log) == len(ref_out) == m["n_forwards"]
    # forwards before the first eviction, and the first evicting forward itself, saw an untouched cache
    n_before = (m["config"]["budget"] + 2) if m["mode"] == "decoding" else 3
    for f in range(min(n_before, len(ref_out))):
        assert out_close(model.outputs_log[f], ref_out[f]), f
    assert len(cache.evictions) == int(g["kinds"].shape[0])


@pytest.mark.parametrize("name", _tie_cases())
def test_tie_goldens_every_victim_is_a_valid_outcome(name):
    """Step by step on the fixture's own streams: the oracle is re-seeded from the bank's state before every forward, so each
    decision is judged on its own; every victim must be a valid outcome of the reference's rule on the rows it saw, and outputs match."""
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import geometry
    from oracle import easykv_oracle as O
    g = load_golden(name)
    m = g["meta"]
    d = m["dims"]
    L, Hq, H, D = d["L"], d["Hq"], d["H"], d["D"]
    qs, ks, vs = g["streams"]
    cap = Capture()
    n_tied = n_total = 0

    def judge(bank, plan_kw, q, k, v, W, off, k1, kk, sink):
        nonlocal n_tied, n_total
        kord, vord = (t.float().cpu() for t in bank.ordered_kv())
        state = [t[:, :, :W].cpu().clone() for t in (bank.score_sum, bank.score_sq, bank.score_cnt)]
        out, ids = bank.attend(StepPlan(**plan_kw), q.cuda(), k.cuda(), v.cuda())
        for l in range(L):
            st = O.LayerState(k=kord[l:l + 1], v=vord[l:l + 1], s=state[0][l], q=state[1][l], c=state[2][l])
            o_ref, _ = O.layer_step(st, q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(), O.StepPlan(**plan_kw))
            assert out_close(out[l].float().cpu(), o_ref[0])
            if not plan_kw["evict"]:
                continue
            std, mean = O.roco_std(cap.s, cap.q, cap.c, sink), cap.s / cap.c
            for h in range(H):
                forced, pool, need = feasible_classes(std[h], k1)
                got = [int(x) - off for x in ids[l, h].cpu().tolist()]
                assert valid_victims(got, mean[h], forced, pool, need, kk), (name, l, h, got)
                n_total += 1
                n_tied += int(len(pool) != need)

    O.SELECT_HOOK = cap
    try:
        if m["mode"] == "decoding":
            P, budget = m["length"], m["config"]["budget"]
            bank = KVBank(L, Hq, H, D, cap=P + budget + 1)
            bank.load_rows(ks[:, :, :P].cuda(), vs[:, :, :P].cuda())
            bank.state_init(budget + 1, 0)
            for i in range(m["n_forwards"] - 1):
                t = P + i
                evict = (bank.n_slots[0] + 1 - P) > budget
                judge(bank, dict(policy="roco", phase="decode", evict=evict, score_off=P, budget=budget), qs[:, :, t:t + 1].contiguous(),
                      ks[:, :, t:t + 1].contiguous(), vs[:, :, t:t + 1].contiguous(), budget + 1, P, budget - int(budget * 0.3), 1, 0)
        else:
            s, length = m["stride"], m["length"]
            budget_p, idx, r_idx = geometry("encoding", length, m["config"]["budget"], s)
            recent, sink = int(budget_p * m["config"].get("recent_ratio", 0.1)), m["config"].get("temp_length", 4)
            bank = KVBank(L, Hq, H, D, cap=idx + s)
            bank.load_rows(ks[:, :, :r_idx].cuda(), vs[:, :, :r_idx].cuda())
            bank.state_init(idx + s, 2, s)
            for t in range(r_idx, length, s):
                t_now = bank.n_slots[0] + s
                kw = dict(policy="roco", phase="prefill", accumulate=t_now > idx, evict=t_now > idx, budget=budget_p, recent=recent, sink=sink, stride=s)
                judge(bank, kw, qs[:, :, t:t + s].contiguous(), ks[:, :, t:t + s].contiguous(), vs[:, :, t:t + s].contiguous(), idx + s, 0,
                      max(budget_p - recent - sink, s), s, sink)
    finally:
        O.SELECT_HOOK = None
    assert n_total > 0 and n_tied > 0, "the fixture was meant to put k1 inside a tied class"
