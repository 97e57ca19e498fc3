"""The logits-resident scored chunk step (easykv_amd/csrc/ekv_attn_resident.inc, round 6): ONE launch per step for 9..64 GQA-folded
query rows against at most 1280 keys (at most 32 rows: 2560 keys) — the logits of a head stay in the register file, K and V are read once, the scorer runs in the
same workgroup — against the two-pass step of the wide-block kernel it replaces there (``two_pass = 1`` forces that one: one pass over
K and V, K-only column-sum pass, scorer as its tail) on a twin bank over several consecutive steps: the same evicted ids, slot map and
count rows, score rows and outputs to rounding (the row statistics are formed in a different order: exact maximum and one sum here,
a running maximum with rescaling there).  The two-pass step is pinned to the oracle / the reference's fixtures by
tests/test_hip_prefill_parity.py, tests/test_hip_fullsize_configs.py (whose configs[2] test now runs on this kernel), tests/test_hip_lockstep.py
and tests/test_hip_wide_kernel.py; the second test here compares this kernel with the oracle directly.

Reference: attention easykv/llama_patch.py:198-222, mistral_patch.py:144-169; accumulate easykv/easykv.py:443-457, select :462-490,
compaction :465-490 / :56-82."""
import pytest
import torch

from tests.test_hip_wide_tail import _bank

pytestmark = pytest.mark.gpu

SHAPES = [
    # hq, h, n, t_prev, policy
    (8, 2, 16, 1232, "roco"),         # configs[2]-shaped: Mistral GQA x4, 64 folded rows, T = 1248
    (8, 2, 16, 1232, "h2o_head"),
    (4, 4, 64, 1216, "roco"),         # the widest step it takes: 64 rows x 1280 keys, ten full tiles
    (4, 4, 40, 5, "roco"),            # a tiny cache: one tile, sentinels inside the feasible set
    (4, 2, 20, 700, "roco"),          # GQA x2, 40 rows; T = 720 ends inside a tile
    (8, 2, 9, 130, "h2o_head"),       # GQA x4, 36 rows; the chunk straddles a tile boundary (keys 130 .. 138 of tiles 1 / 2)
    (4, 4, 48, 80, "roco"),           # the chunk's own rows start in tile 0 and end in tile 1
    (4, 4, 33, 0, "roco"),            # empty cache: the first chunk (nothing to evict)
    (4, 2, 17, 1000, "h2o_head"),     # GQA x2, 34 rows
    (4, 4, 32, 1000, "roco"),         # 32 rows: the second query wave of every pair is empty
    (8, 2, 4, 700, "h2o_head"),       # GQA x4, 16 rows
    (4, 4, 9, 300, "roco"),           # 9 rows, the fewest it takes (8 and fewer: the logits-in-LDS kernel)
    (8, 2, 8, 2056, "roco"),          # LONG shape (<= 32 rows, T <= 2560): Mistral GQA x4 at stride 8, budget 0.5 of 4096: T = 2064
    (4, 4, 32, 2500, "h2o_head"),     # ... 32 rows, T = 2532 ends inside the last of 20 tiles
    (4, 4, 16, 1270, "roco"),         # ... T = 1286, just past the short shape: tile 10 (the second group's) holds 6 keys
    (16, 2, 4, 2000, "roco"),         # ... GQA x8, 32 rows
    (4, 2, 5, 1400, "h2o_head"),      # ... GQA x2, 10 rows
    (16, 2, 8, 600, "roco"),          # GQA x8, 64 rows: a query's heads span both half-waves
    (16, 2, 5, 1275, "h2o_head"),     # GQA x8, 40 rows, T = 1280
]


@pytest.mark.parametrize("hq,h,n,t_prev,policy", SHAPES)
def test_resident_step_equals_the_two_pass_step(hq, h, n, t_prev, policy):
    from easykv_amd import StepPlan
    L, d, T, steps = 3, 128, t_prev + n, 6
    g = torch.Generator().manual_seed(11 * hq + 5 * n + t_prev)
    k = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    v = torch.randn(L, h, t_prev + n * steps, d, generator=g).half()
    q = torch.randn(L, hq, n * steps, d, generator=g).half()
    a = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 5)
    b = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 5)
    sink = 4 if t_prev > 100 else 0
    kw = dict(policy=policy, phase="prefill", accumulate=True, evict=t_prev > 0, budget=T, recent=int(T * 0.1), sink=sink, stride=n)
    plan_a, plan_b = StepPlan(**kw), StepPlan(two_pass=1, **kw)
    ia_, ib_ = a.step_info(plan_a, n), b.step_info(plan_b, n)
    assert (ia_["fused"], ia_["n_launches"]) == (1, 1), ia_          # the whole step in one launch
    assert ib_["n_launches"] in (2, 3), ib_                          # one pass + column-sum pass with the scorer as its tail (split heads: + scorer)
    for s in range(steps if t_prev > 0 else 1):
        sl = slice(t_prev + s * n, t_prev + (s + 1) * n)
        qs, ks, vs = q[:, :, s * n:(s + 1) * n].cuda().contiguous(), k[:, :, sl].cuda().contiguous(), v[:, :, sl].cuda().contiguous()
        oa, ia = a.attend(plan_a, qs, ks, vs)
        ob, ib = b.attend(plan_b, qs, ks, vs)
        assert (oa.float() - ob.float()).abs().max().item() <= 1e-3, s
        if plan_a.evict:
            assert torch.equal(ia, ib), (s, (ia != ib).nonzero()[:4].tolist())
        assert a.n_slots == b.n_slots
        assert torch.equal(a.slot_of_pos, b.slot_of_pos), s
        rows = a.slot_of_pos[:, :, :a.n_slots[0]].long().unsqueeze(-1).expand(-1, -1, -1, d)       # the chunk's own rows were appended to the same slots
        assert torch.equal(a.k.gather(2, rows), b.k.gather(2, rows)) and torch.equal(a.v.gather(2, rows), b.v.gather(2, rows)), s
        assert torch.equal(a.score_cnt, b.score_cnt), s
        torch.testing.assert_close(a.score_sum, b.score_sum, rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(a.score_sq, b.score_sq, rtol=4e-5, atol=1e-7)


@pytest.mark.parametrize("hq,h,n,t_prev", [(8, 2, 16, 1232), (4, 4, 50, 300), (4, 2, 17, 1000), (16, 2, 8, 500), (8, 2, 8, 900), (4, 4, 12, 1268), (8, 2, 8, 2056), (4, 4, 20, 1500)])
def test_resident_step_against_the_oracle(hq, h, n, t_prev):
    """Output rows, column sums (through the score rows of a fresh state) and the victims of one step against the oracle's chunk step on
    the same scattered cache."""
    from easykv_amd import StepPlan
    from oracle import easykv_oracle as O
    from tests.golden_util import out_close
    L, d, T = 2, 128, t_prev + n
    rep = hq // h
    g = torch.Generator().manual_seed(3 * hq + n + t_prev)
    k = torch.randn(L, h, T, d, generator=g).half()
    v = torch.randn(L, h, T, d, generator=g).half()
    q = torch.randn(L, hq, n, d, generator=g).half()
    a = _bank(L, hq, h, d, T, n, t_prev, k, v, False, 5)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=False, budget=T, recent=int(T * 0.1), sink=4, stride=n)
    assert a.step_info(plan, n)["n_launches"] == 1
    out, _ = a.attend(plan, q.cuda(), k[:, :, t_prev:].cuda().contiguous(), v[:, :, t_prev:].cuda().contiguous())
    for layer in range(L):
        o_ref, p = O.attention_core(q[layer:layer + 1].float(), k[layer:layer + 1].float(), v[layer:layer + 1].float(),
                                    O.causal_chunk_mask(n, T, torch.float32))
        assert out_close(out[layer:layer + 1].cpu(), o_ref)
        pbar = p.reshape(1, h, rep, n, T).mean(2)                     # GQA fold (easykv/easykv.py:173-186), then the column sums (:443-457)
        torch.testing.assert_close(a.score_sum[layer, :, :T].cpu(), pbar.sum(2)[0], rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(a.score_sq[layer, :, :T].cpu(), (pbar ** 2).sum(2)[0], rtol=4e-5, atol=1e-7)
