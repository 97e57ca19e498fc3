"""The scorer of a two-pass wide chunk step as the TAIL of the column-sum pass (easykv_amd/csrc/ekv_wide_tail.h, round 5) against
the stand-alone scorer launch it replaces (``phases = 1`` then ``phases = 2``: the same attention launches, then
ekv_score_select_kernel) on a twin bank: evicted ids, slot map and score rows must be EQUAL bit for bit — same column sums, same
order of the sums, exact selects on both sides — over several consecutive steps, plain and RoPE-on-read keys, GQA, unsplit heads
(the head's own workgroup scores it) and key-range splits / RoPE-on-read (which keep the stand-alone scorer: equal trivially, the
launch count says which form ran); and the flush of a deferred step whose column-sum pass runs UNSPLIT over all layers although
the one-layer calls split their key ranges (the tail folds `n_stat_parts` row-statistics partials).  The stand-alone scorer is pinned to the oracle / the reference's fixtures by
tests/test_hip_prefill_parity.py, tests/test_hip_fullsize_configs.py and tests/test_hip_wide_kernel.py.

Reference: accumulate easykv/easykv.py:443-457, select :462-490, compaction :465-490 / :56-82."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # d, hq, h, n, t_prev, n_split, policy, streaming
    (128, 4, 4, 96, 1200, 1, "roco", False),        # configs[3]-shaped: 96 rows, unsplit
    (128, 4, 4, 96, 5002, 1, "roco", False),        # ... at its full width (W = 5098: 20 columns per thread)
    (128, 4, 4, 96, 1200, 0, "roco", False),        # library-chosen splits (few heads: several workgroups per head, stand-alone scorer)
    (128, 4, 4, 96, 2100, 4, "roco", False),
    (128, 8, 2, 16, 1232, 1, "h2o_head", False),    # configs[2]-shaped: Mistral GQA x4, 64 folded rows (2 x 2 waves)
    (128, 8, 2, 16, 1232, 2, "roco", False),
    (64, 4, 4, 64, 700, 1, "roco", False),
    (128, 4, 4, 96, 900, 1, "roco", True),          # configs[4]-shaped: RoPE-on-read (keeps the stand-alone scorer: equal trivially, 3 launches)
    (128, 4, 4, 96, 900, 3, "h2o_head", True),
    (128, 16, 2, 12, 400, 1, "roco", False),        # GQA x8 (run-time fold)
    (128, 4, 4, 40, 5, 1, "roco", False),           # a tiny cache: sentinels inside the feasible set
]


def _bank(L, hq, h, d, T, n, t_prev, k, v, streaming, seed):
    from easykv_amd import KVBank
    g = torch.Generator().manual_seed(seed)
    bank = KVBank(L, hq, h, d, cap=T + 64)
    if streaming:
        from easykv_amd.api import rope_tables
        bank.set_rope(*rope_tables(T + 64, d))
    if t_prev:
        bank.load_rows(k[:, :, :t_prev].cuda(), v[:, :, :t_prev].cuda())
        perm = torch.argsort(torch.rand(L, h, t_prev, generator=g), dim=-1).int().cuda()
        kk, vv = bank.k.clone(), bank.v.clone()
        idx = perm.long().unsqueeze(-1).expand(-1, -1, -1, d)
        bank.k[:, :, :t_prev].scatter_(2, idx, kk[:, :, :t_prev])
        bank.v[:, :, :t_prev].scatter_(2, idx, vv[:, :, :t_prev])
        bank.slot_of_pos[:, :, :t_prev] = perm
    bank.state_init(T, 2, n)
    return bank


@pytest.mark.parametrize("d,hq,h,n,t_prev,n_split,policy,streaming", SHAPES)
def test_tail_equals_the_stand_alone_scorer(d, hq, h, n, t_prev, n_split, policy, streaming):
    from easykv_amd import StepPlan
    L, T, steps = 3, t_prev + n, 6
    g = torch.Generator().manual_seed(7 * d + 13 * hq + n + t_prev)
    k = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    v = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    q = torch.randn(L, hq, n * steps, d, generator=g).half()
    a = _bank(L, hq, h, d, T, n, t_prev, k, v, streaming, 5)
    b = _bank(L, hq, h, d, T, n, t_prev, k, v, streaming, 5)
    budget_p = t_prev + n
    sink = 4 if t_prev > 100 else 0
    plan = StepPlan(policy=policy, phase="prefill", accumulate=True, evict=t_prev > 0, budget=budget_p, recent=int(budget_p * 0.1), sink=sink, stride=n,
                    n_split=n_split, two_pass=1, streaming=streaming)
    info = a.step_info(plan, n)
    assert info["wide"] == 1 and info["two_pass"] == 1
    if not streaming and info["n_split"] * info["n_col_parts"] == 1:
        assert info["n_launches"] == 2, info          # one pass + column-sum pass with the scorer as its tail
    else:
        assert info["n_launches"] == 3, info          # RoPE-on-read (two workgroups per CU) and split heads keep the stand-alone scorer
    for s in range(steps if t_prev > 0 else 1):
        sl = slice(t_prev + s * n, t_prev + (s + 1) * n)
        qs, ks, vs = q[:, :, s * n:(s + 1) * n].cuda().contiguous(), k[:, :, sl].cuda().contiguous(), v[:, :, sl].cuda().contiguous()
        oa, ia = a.attend(plan, qs, ks, vs)                                   # whole step: the tail scores
        ob = torch.empty_like(oa)
        ib = torch.empty(L, h, n, dtype=torch.int32, device="cuda") if plan.evict else None
        b.attend(plan, qs, ks, vs, out=ob, evict_ids=ib, phases=1)            # attention launches only
        b.attend(plan, qs, ks, vs, out=ob, evict_ids=ib, phases=2)            # fold + stand-alone scorer
        assert torch.equal(oa, ob), s
        if plan.evict:
            assert torch.equal(ia, ib), (s, (ia != ib).nonzero()[:4].tolist())
        assert a.n_slots == b.n_slots
        assert torch.equal(a.slot_of_pos, b.slot_of_pos), s
        assert torch.equal(a.score_sum, b.score_sum) and torch.equal(a.score_sq, b.score_sq) and torch.equal(a.score_cnt, b.score_cnt), s


def test_tail_on_a_broad_score_distribution_and_exact_ties():
    """Keys with log-normal norms (score keys spread over decades: crowded histogram bins, refinement levels) and a block of IDENTICAL
    cached rows (exact ties in mean and std: the tie rule 'lower position first' decides) — still equal to the stand-alone scorer."""
    from easykv_amd import StepPlan
    L, hq, h, d, n, t_prev, steps = 2, 4, 4, 128, 96, 3000, 5
    g = torch.Generator().manual_seed(99)
    T = t_prev + n
    k = torch.randn(L, h, t_prev + n * steps, d, generator=g)
    k = (k * torch.exp(1.2 * torch.randn(L, h, k.shape[2], 1, generator=g))).half()
    k[:, :, 500:900] = k[:, :, 500:501]                 # 400 identical rows
    v = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    q = torch.randn(L, hq, n * steps, d, generator=g).half()
    a = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 6)
    b = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 6)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=T, recent=int(T * 0.1), sink=4, stride=n, n_split=1, two_pass=1)
    for s in range(steps):
        sl = slice(t_prev + s * n, t_prev + (s + 1) * n)
        qs, ks, vs = q[:, :, s * n:(s + 1) * n].cuda().contiguous(), k[:, :, sl].cuda().contiguous(), v[:, :, sl].cuda().contiguous()
        oa, ia = a.attend(plan, qs, ks, vs)
        ob, ib = torch.empty_like(oa), torch.empty_like(ia)
        b.attend(plan, qs, ks, vs, out=ob, evict_ids=ib, phases=1)
        b.attend(plan, qs, ks, vs, out=ob, evict_ids=ib, phases=2)
        assert torch.equal(ia, ib), s
        assert torch.equal(a.slot_of_pos, b.slot_of_pos) and torch.equal(a.score_sum, b.score_sum) and torch.equal(a.score_cnt, b.score_cnt), s


def test_launches_of_more_than_one_workgroup_per_cu_keep_the_four_wave_tiles():
    """Round 5 runs launches of at most 256 workgroups (a layer-per-call model) on 128-key tiles with 4 x 2 waves; everything else in this
    file is such a launch.  288 heads (9 layers x 32) take the 64-key, 4-wave shape the 32-layer bench launches run on: whole step (tail)
    against attention launches + stand-alone scorer on a twin bank, and the attention output of one layer against the oracle."""
    from easykv_amd import StepPlan
    from oracle import easykv_oracle as O
    from tests.golden_util import out_close
    L, hq, h, d, n, t_prev, steps = 9, 32, 32, 128, 96, 700, 3
    T = t_prev + n
    g = torch.Generator().manual_seed(4242)
    k = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    v = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    q = torch.randn(L, hq, n * steps, d, generator=g).half()
    a = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 8)
    b = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 8)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=T, recent=int(T * 0.1), sink=4, stride=n, two_pass=1)
    info = a.step_info(plan, n)
    assert info["wide"] == 1 and info["n_split"] == 1 and info["n_launches"] == 2
    for s in range(steps):
        sl = slice(t_prev + s * n, t_prev + (s + 1) * n)
        qs, ks, vs = q[:, :, s * n:(s + 1) * n].cuda().contiguous(), k[:, :, sl].cuda().contiguous(), v[:, :, sl].cuda().contiguous()
        if s == 0:      # the oracle's attention over the same (scattered) cache: layer 4
            kk, vv = a.ordered_kv(4, 1)
            o_ref, _ = O.attention_core(q[4:5, :, :n].float(), torch.cat([kk.float().cpu(), k[4:5, :, sl].float()], 2),
                                        torch.cat([vv.float().cpu(), v[4:5, :, sl].float()], 2), O.causal_chunk_mask(n, T, torch.float32))
        oa, ia = a.attend(plan, qs, ks, vs)
        ob, ib = torch.empty_like(oa), torch.empty_like(ia)
        b.attend(plan, qs, ks, vs, out=ob, evict_ids=ib, phases=1)
        b.attend(plan, qs, ks, vs, out=ob, evict_ids=ib, phases=2)
        if s == 0:
            assert out_close(oa[4].float().cpu(), o_ref[0]), float((oa[4].float().cpu() - o_ref[0]).abs().max())
        assert torch.equal(oa, ob) and torch.equal(ia, ib), s
        assert torch.equal(a.slot_of_pos, b.slot_of_pos) and torch.equal(a.score_sum, b.score_sum) and torch.equal(a.score_cnt, b.score_cnt), s


@pytest.mark.parametrize("streaming", [False, True], ids=["plain", "rope"])
def test_deferred_flush_with_512_pairs_runs_unsplit_and_equals_the_whole_steps(streaming):
    """a deferred flush over >= 512 (layer, head) pairs runs its column-sum pass UNSPLIT with the scorer as its tail while the
    one-layer calls of the one pass used key-range splits (n_stat_parts != n_split: the pass folds the row-statistics partials of every
    split) — plain keys only: a RoPE-on-read flush keeps the split and the stand-alone scorer.  Both forms must equal the immediate
    per-layer steps bit for bit: outputs, evicted sets, slot maps, score rows."""
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import rope_tables
    L, hq, h, d, n, t0, steps = 16, 32, 32, 128, 96, 600, 2
    g = torch.Generator().manual_seed(31 + int(streaming))
    k0, v0 = torch.randn(L, h, t0, d, generator=g).half().cuda(), torch.randn(L, h, t0, d, generator=g).half().cuda()
    qs = [torch.randn(L, hq, n, d, generator=g).half().cuda() for _ in range(steps)]
    ks = [torch.randn(L, h, n, d, generator=g).half().cuda() for _ in range(steps)]
    vs = [torch.randn(L, h, n, d, generator=g).half().cuda() for _ in range(steps)]
    res = {}
    for mode in ("whole", "deferred"):
        bank = KVBank(L, hq, h, d, cap=t0 + n)
        if streaming:
            bank.set_rope(*rope_tables(bank.cap + 8, d))
        bank.load_rows(k0, v0)
        bank.state_init(t0 + n, 2, n)
        outs, idl = [], []
        for i in range(steps):
            plan = StepPlan(policy="roco", phase="prefill", evict=True, accumulate=True, budget=t0 + n, recent=40, sink=4, stride=n,
                            streaming=streaming, n_split=2, two_pass=1)
            out = torch.empty(L, hq, n, d, dtype=torch.float16, device="cuda")
            if mode == "whole":
                ids = torch.full((L, h, n), -1, dtype=torch.int32, device="cuda")
                for l in range(L):
                    bank.attend(plan, qs[i][l:l + 1], ks[i][l:l + 1], vs[i][l:l + 1], layer_begin=l, out=out[l:l + 1], evict_ids=ids[l:l + 1])
            else:
                for l in range(L):
                    bank.attend(plan, qs[i][l:l + 1], ks[i][l:l + 1], vs[i][l:l + 1], layer_begin=l, out=out[l:l + 1], defer=True)
                if i == 0:      # the flush call's launches: column-sum pass (+ the stand-alone scorer unless it runs as the pass's tail)
                    st = bank._defer["st"]
                    st.layer_begin, st.layer_count, st.defer_index, st.phases = 0, L, 0, 8
                    info = (__import__("ctypes").c_int32 * 9)()
                    assert bank.lib.ekv_step_info(__import__("ctypes").byref(bank._bank), __import__("ctypes").byref(st), info, 9) == 0
                    assert info[0] == 2 and info[8] == (2 if streaming else 1), list(info)
                ids = bank.flush().clone()
            outs.append(out)
            idl.append(torch.sort(ids, dim=-1)[0])
        torch.cuda.synchronize()
        res[mode] = (torch.stack(outs), torch.stack(idl), bank.slot_of_pos[:, :, :t0].clone(), bank.score_sum[:, :, :t0].clone(),
                     bank.score_sq[:, :, :t0].clone(), bank.score_cnt[:, :, :t0].clone())
    for a, b in zip(res["whole"], res["deferred"]):
        assert torch.equal(a, b)
