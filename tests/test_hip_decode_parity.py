"""GPU parity of the fused decode step (through the C ABI) against the committed golden vectors
(produced by the real reference) and against the CPU oracle on fresh seeded inputs.

Bar: eviction index sets bit-identical; attention outputs within 1e-3 (fp16 outputs, north_star)."""
import numpy as np
import pytest
import torch

from tests.golden_util import golden_names, load_golden, split_ids, split_outputs, out_close

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-3   # north_star: within 1e-3 on fp16 outputs — flat absolute bound, rtol = 0


def _replay_decoding(g, n_split=0):
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    m = g["meta"]
    d = m["dims"]
    L, Hq, H, D = d["L"], d["Hq"], d["H"], d["D"]
    P, budget, policy = m["length"], m["config"]["budget"], m["config"]["kv_policy"]
    max_new = m["config"]["max_new_tokens"]
    qs, ks, vs = (x.cuda() for x in g["streams"])
    bank = KVBank(L, Hq, H, D, cap=P + max(budget, max_new) + 1)
    if m["streaming"]:
        cos, sin = O.rope_tables(P + budget + 8, D)
        bank.set_rope(cos, sin)
    bank.load_rows(ks[:, :, :P], vs[:, :, :P])
    bank.state_init(budget + 1, 0)
    known = policy in ("roco", "h2o_head", "tova", "recency", "random", "full")
    ids_log, outs = [], []
    for i in range(m["n_forwards"] - 1):     # the reference's decode forwards (fewer than max_new_tokens when it met an EOS)
        t = P + i
        gen = bank.n_slots[0] + 1 - P
        evict = gen > budget and policy != "full" and known   # unknown strings evict nothing
        plan = StepPlan(policy=policy, phase="decode", evict=evict, score_off=P, budget=budget,
                        streaming=m["streaming"], n_split=n_split)
        if policy == "recency" and evict:
            plan.range_start = P      # oldest generated slot (easykv/easykv.py:343-347)
        out, ids = bank.attend(plan, qs[:, :, t:t + 1].contiguous(), ks[:, :, t:t + 1].contiguous(), vs[:, :, t:t + 1].contiguous())
        outs.append(out.float().cpu())
        if ids is not None:
            ids_log.append(ids.cpu().numpy())
    return ids_log, outs, bank


def _decoding_cases():
    out = []
    for n in golden_names():
        m = load_golden(n)["meta"]
        # ('random' draws its victim on the host, in the driver: replayed end to end by tests/test_hip_generate_parity.py)
        if m["mode"] == "decoding" and m["tie_free"] and m["config"]["kv_policy"] != "random":
            out.append(n)
    return out


@pytest.mark.parametrize("name", _decoding_cases())
@pytest.mark.parametrize("n_split", [0, 3])
def test_decode_matches_reference_golden(name, n_split):
    g = load_golden(name)
    ids_log, outs, bank = _replay_decoding(g, n_split)
    ref_out = split_outputs(g)[1:]          # forward 0 is the prompt prefill
    assert len(outs) == len(ref_out)
    for a, b in zip(outs, ref_out):
        assert out_close(a, b, OUT_TOL), float((a - b).abs().max())
    if g["meta"]["config"]["kv_policy"] == "recency":
        ref = [np.broadcast_to(np.array(r[0]), ids_log[0].shape) for r in g["ranges"]]
    else:
        ref = split_ids(g)
    assert len(ids_log) == len(ref)
    for step, (a, b) in enumerate(zip(ids_log, ref)):
        assert np.array_equal(a, b), f"eviction ids differ at eviction step {step}"
    k_ord, _ = bank.ordered_kv()
    kept = int(g["meta"]["printed"].split("(")[1].split("/")[0])     # the reference's own printed budget line
    assert k_ord.shape[2] == g["meta"]["length"] + kept


@pytest.mark.parametrize("hq,h,d,policy,stream", [(8, 8, 64, "roco", False), (16, 4, 128, "roco", False), (16, 2, 32, "h2o_head", True),
                                                  (8, 8, 128, "tova", False), (32, 4, 64, "roco", True)])
def test_eight_wave_fused_kernel_for_launches_with_one_or_two_heads_per_cu(hq, h, d, policy, stream):
    """256..512 KV heads in one launch (e.g. Mistral: 8 KV heads x 32 layers) run the fused decode step with 8-wave
    workgroups, one per head.  Same trajectory as the split path (attention kernel + decode scorer), layer by layer
    against the oracle for the first layers."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    L = 256 // h
    budget, steps, P = 150, 12, 5
    g = torch.Generator().manual_seed(hq * 7 + d)
    T0 = P + budget
    k0, v0 = torch.randn(L, h, T0, d, generator=g).half(), torch.randn(L, h, T0, d, generator=g).half()
    warm = torch.rand(L, h, budget + 1, generator=g) * 1e-3
    banks = {}
    cos = sin = None
    if stream:
        cos, sin = O.rope_tables(T0 + 16, d)
    for name in ("fused", "split"):
        b = KVBank(L, hq, h, d, cap=T0 + 1)
        if stream:
            b.set_rope(cos, sin)
        b.load_rows(k0.cuda(), v0.cuda())
        b.state_init(budget + 1, 0)
        b.score_sum[:, :, :budget + 1] += warm.cuda()
        b.score_sq[:, :, :budget + 1] += (warm ** 2).cuda()
        banks[name] = b
    kw = dict(policy=policy, phase="decode", evict=True, score_off=P, budget=budget, streaming=stream)
    assert banks["fused"].step_plan(StepPlan(**kw), 1) == (1, True)          # auto plan: unsplit and fused
    n_check = 2
    sts = []
    for l in range(n_check):
        st = O.LayerState(k=k0[l:l + 1].float(), v=v0[l:l + 1].float())
        st.s, st.q, st.c = O.init_state_decoding((h,), budget)
        st.s += warm[l]
        st.q += warm[l] ** 2
        sts.append(st)
    for i in range(steps):
        q, k, v = (torch.randn(L, n, 1, d, generator=g).half() for n in (hq, h, h))
        o_f, i_f = banks["fused"].attend(StepPlan(**kw), q.cuda(), k.cuda(), v.cuda())
        o_s, i_s = banks["split"].attend(StepPlan(n_split=2, **kw), q.cuda(), k.cuda(), v.cuda())
        assert torch.equal(i_f, i_s), i
        assert out_close(o_f.float(), o_s.float())
        for l in range(n_check):
            o_ref, ids_ref = O.layer_step(sts[l], q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(), O.StepPlan(**kw), cos, sin)
            assert out_close(o_f[l].float().cpu(), o_ref[0])
    assert torch.equal(banks["fused"].slot_of_pos, banks["split"].slot_of_pos)
    assert torch.allclose(banks["fused"].score_sum, banks["split"].score_sum, rtol=1e-6, atol=0)


@pytest.mark.parametrize("hq,h,d,L,policy,holes", [(8, 8, 128, 8, "roco", 37), (8, 4, 64, 4, "roco", 5), (16, 2, 32, 3, "h2o_head", 90),
                                                   (4, 4, 128, 64, "roco", 1), (32, 4, 128, 2, "tova", 64), (8, 8, 128, 40, "recency", 12)])
def test_physical_order_stream_with_a_scattered_slot_map(hq, h, d, L, policy, holes):
    """The one-launch decode step streams the K/V rows in PHYSICAL order and masks the dead ones (free rows, the row being
    appended, the padding past the extent).  Worst case here: the live rows are a random subset of the bank's rows in a random
    order, the free rows in between hold NaN / inf garbage, and the extent is unknown (= cap).  Same trajectory as the
    position-ordered split path and the oracle; the natural case (extent tracked by KVBank, holes left by a chunk phase) follows."""
    from easykv_amd import KVBank, StepPlan
    from oracle import easykv_oracle as O
    budget, steps, P = 150, 24, (0 if policy == "recency" else 7)
    g = torch.Generator().manual_seed(hq * 11 + d + holes)
    T0 = P + budget
    cap = T0 + 1 + holes
    k0, v0 = torch.randn(L, h, T0, d, generator=g).half(), torch.randn(L, h, T0, d, generator=g).half()
    warm = torch.rand(L, h, budget + 1, generator=g) * 1e-3
    banks = {}
    for name in ("fused", "split"):
        b = KVBank(L, hq, h, d, cap=cap)
        # scatter: logical position j lives in physical row perm[j]; rows perm[T0:] are free and poisoned
        gp = torch.Generator().manual_seed(99)
        perm = torch.stack([torch.randperm(b.cap, generator=gp) for _ in range(L * h)]).view(L, h, b.cap).int().cuda()
        b.slot_of_pos.copy_(perm)
        b.k.fill_(float("nan"))
        b.v.fill_(float("inf"))
        b.load_rows(k0.cuda(), v0.cuda())          # goes through the slot map
        b.extent = [b.cap] * L                      # the free list is not in library order any more: extent unknown
        if policy != "recency":
            b.state_init(budget + 1, 0)
            b.score_sum[:, :, :budget + 1] += warm.cuda()
            b.score_sq[:, :, :budget + 1] += (warm ** 2).cuda()
        banks[name] = b
    kw = dict(policy=policy, phase="decode", evict=True, score_off=P, budget=budget)
    if policy == "recency":
        kw["range_start"] = 0
    assert banks["fused"].step_plan(StepPlan(n_split=1, **kw), 1) == (1, True)
    n_check = min(2, L)
    sts = []
    for l in range(n_check):
        st = O.LayerState(k=k0[l:l + 1].float(), v=v0[l:l + 1].float())
        if policy != "recency":
            st.s, st.q, st.c = O.init_state_decoding((h,), budget)
            st.s += warm[l]
            st.q += warm[l] ** 2
        sts.append(st)
    for i in range(steps):
        q, k, v = (torch.randn(L, n, 1, d, generator=g).half() for n in (hq, h, h))
        o_f, i_f = banks["fused"].attend(StepPlan(n_split=1, **kw), q.cuda(), k.cuda(), v.cuda())
        o_s, i_s = banks["split"].attend(StepPlan(n_split=2, **kw), q.cuda(), k.cuda(), v.cuda())
        assert torch.equal(i_f, i_s), i
        assert torch.isfinite(o_f.float()).all()
        assert out_close(o_f.float(), o_s.float())
        for l in range(n_check):
            o_ref, ids_ref = O.layer_step(sts[l], q[l:l + 1].float(), k[l:l + 1].float(), v[l:l + 1].float(), O.StepPlan(**kw))
            assert out_close(o_f[l].float().cpu(), o_ref[0])
    assert torch.equal(banks["fused"].slot_of_pos, banks["split"].slot_of_pos)
    if policy != "recency":
        assert torch.equal(banks["fused"].score_sum, banks["split"].score_sum)     # same summation order: bit-identical
    kf, vf = banks["fused"].ordered_kv()
    ks, vs = banks["split"].ordered_kv()
    assert torch.equal(kf, ks) and torch.equal(vf, vs)


def test_physical_order_stream_after_a_chunk_phase_leaves_holes():
    """auto-mode shape: a strided chunk phase (8 victims per step) followed by single-token decode — the decode steps run
    with 7 permanent holes below KVBank's tracked extent.  Fused (physical order) vs split (position order), 200 steps."""
    from easykv_amd import KVBank, StepPlan
    L, hq, h, d, stride, idx = 16, 8, 8, 128, 8, 120
    g = torch.Generator().manual_seed(5)
    banks = {n: KVBank(L, hq, h, d, cap=idx + stride + 1) for n in ("fused", "split")}
    k0, v0 = torch.randn(L, h, idx, d, generator=g).half(), torch.randn(L, h, idx, d, generator=g).half()
    for b in banks.values():
        b.load_rows(k0.cuda(), v0.cuda())
        b.state_init(idx + stride, 2, stride)
    pre = dict(policy="roco", phase="prefill", accumulate=True, evict=True, budget=idx + stride, recent=12, sink=4, stride=stride)
    for i in range(6):
        q, k, v = (torch.randn(L, n, stride, d, generator=g).half() for n in (hq, h, h))
        outs = [b.attend(StepPlan(**pre), q.cuda(), k.cuda(), v.cuda()) for b in banks.values()]
        assert torch.equal(outs[0][1], outs[1][1])
    assert banks["fused"].extent == [idx + stride] * L and banks["fused"].n_slots == [idx] * L
    dec = dict(policy="roco", phase="decode", evict=True, score_off=0, budget=idx + stride)
    assert banks["fused"].step_plan(StepPlan(n_split=1, **dec), 1) == (1, True)
    for i in range(200):
        q, k, v = (torch.randn(L, n, 1, d, generator=g).half() for n in (hq, h, h))
        o_f, i_f = banks["fused"].attend(StepPlan(n_split=1, **dec), q.cuda(), k.cuda(), v.cuda())
        o_s, i_s = banks["split"].attend(StepPlan(n_split=2, **dec), q.cuda(), k.cuda(), v.cuda())
        assert torch.equal(i_f, i_s), i
        assert out_close(o_f.float(), o_s.float())
    assert banks["fused"].extent == [idx + stride] * L
    assert torch.equal(banks["fused"].slot_of_pos, banks["split"].slot_of_pos)
    assert torch.equal(banks["fused"].score_sum, banks["split"].score_sum)
