"""Seam hygiene: the cache of the forward in flight is per context, not a process-wide global;
a model whose attention is not routed through the cache fails loudly; unscored range evictions ('recency' / 'random') have no
row-width limit; the documented ctypes stub of INTEGRATION.md §3 binds the library as written."""
import contextlib
import io
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_util import load_golden, split_ids, out_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen(name, model_hook=None, **extra):
    import easykv_amd
    from tests.native_fake_model import NativeFakeModel
    g = load_golden(name)
    m = g["meta"]
    model = NativeFakeModel(*g["streams"], arch=m["arch"], vocab=m.get("vocab", 16))
    if model_hook is not None:
        model_hook(model)
    cfg = dict(m["config"], eos_token_ids=m.get("eos_token_ids", [-1]), _record_evictions=True, **extra)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        res, cache = easykv_amd.generate(model, torch.arange(m["length"]).view(1, -1) % 16, cfg, kv_mode=m["mode"], stride=m["stride"],
                                         return_cache=True)
    return g, res, cache, buf.getvalue().strip()


def _check_ids(g, cache):
    ours = [np.sort(torch.stack(e).cpu().numpy(), axis=-1) for e in cache.evictions]
    ref = split_ids(g)
    assert len(ours) == len(ref)
    for a, b in zip(ours, ref):
        assert np.array_equal(a, b)


def test_nested_generate_inside_a_forward_keeps_both_caches_apart():
    """A second easykv_generate (another model, another cache) started from INSIDE a forward of the first one — the worst
    interleaving a process-wide 'current cache' cannot survive.  Both runs must still equal the reference."""
    from easykv_amd import api
    inner = {}

    def hook(model):
        orig = type(model).__call__
        state = {"n": 0}

        def call(self, *a, **kw):
            state["n"] += 1
            mine = api.active_cache()
            if state["n"] == 5:                     # mid-run: a complete nested generate on a different model
                inner["run"] = _gen("enc_roco_s4")
                assert api.active_cache() is mine   # ... and this forward's cache is the active one again
            return orig(self, *a, **kw)

        model.__class__ = type("Hooked", (type(model),), {"__call__": call})

    g, res, cache, printed = _gen("dec_roco", model_hook=hook)
    assert printed == g["meta"]["printed"] and res == g["meta"]["result"]
    _check_ids(g, cache)
    gi, resi, cachei, printedi = inner["run"]
    assert printedi == gi["meta"]["printed"] and resi == gi["meta"]["result"]
    _check_ids(gi, cachei)
    assert api.active_cache() is None               # nothing is left behind after generate


def test_hf_seam_outside_generate_raises():
    from easykv_amd import hf
    q = torch.zeros(1, 2, 1, 32, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="outside easykv_generate"):
        hf._easykv_attention(None, q, q, q)


def test_model_that_bypasses_the_cache_fails_loudly():
    import easykv_amd
    from tests.native_fake_model import NativeFakeModel
    g = load_golden("dec_roco")

    class Bypass(NativeFakeModel):
        def __call__(self, input_ids, past_key_values=None, position_ids=None, **kw):
            from oracle.fake_model import one_hot_logits
            from types import SimpleNamespace
            return SimpleNamespace(logits=one_hot_logits(position_ids[0].cpu()).to(self.device))   # never calls attend()

    with pytest.raises(RuntimeError, match="attend"):
        easykv_amd.generate(Bypass(*g["streams"]), torch.arange(16).view(1, -1), dict(budget=40, kv_policy="roco", max_new_tokens=4, eos_token_ids=[-1]),
                            kv_mode="decoding")
    with pytest.raises(ValueError, match="batch"):
        easykv_amd.generate(NativeFakeModel(*g["streams"]), torch.zeros(2, 16, dtype=torch.long), dict(budget=40), kv_mode="decoding")


@pytest.mark.parametrize("q_len,k", [(1, 1), (8, 8)])
def test_range_eviction_has_no_row_width_limit(q_len, k):
    """'recency' / 'random' on a 12k-slot cache (the generic scorer's LDS check used to refuse W > ~10k AFTER the
    attention kernel had appended the rows): slot map compacted like the oracle's list delete, outputs = plain attention."""
    from easykv_amd import KVBank, StepPlan
    L, H, D, T0 = 1, 2, 64, 12000
    g = torch.Generator().manual_seed(5)
    k0, v0 = torch.randn(L, H, T0, D, generator=g).half(), torch.randn(L, H, T0, D, generator=g).half()
    bank = KVBank(L, H, H, D, cap=T0 + 64)
    bank.load_rows(k0.cuda(), v0.cuda())
    order = list(range(T0))
    for step in range(3):
        q = torch.randn(L, H, q_len, D, generator=g).half()
        kn, vn = torch.randn(L, H, q_len, D, generator=g).half(), torch.randn(L, H, q_len, D, generator=g).half()
        start = 4 + 1000 * step
        plan = StepPlan(policy="recency", phase="decode" if q_len == 1 else "prefill", accumulate=False, evict=True, stride=k, range_start=start)
        out, ids = bank.attend(plan, q.cuda(), kn.cuda(), vn.cuda())
        assert ids.shape == (L, H, k) and bool((ids.cpu() == torch.arange(start, start + k)).all())
        kk, vv = bank.ordered_kv()
        # reference semantics: append, attend over everything, then delete [start, start+k)  (easykv/easykv.py:105-112)
        order += [T0 + step * q_len + i for i in range(q_len)]
        kn_hist = kn if step == 0 else torch.cat((kn_hist, kn), dim=2)
        v_hist = vn if step == 0 else torch.cat((v_hist, vn), dim=2)
        k_src, v_src = torch.cat((k0, kn_hist), dim=2), torch.cat((v0, v_hist), dim=2)
        idx = torch.tensor(order)
        ka, va = k_src[:, :, idx].float(), v_src[:, :, idx].float()
        w = torch.matmul(q.float(), ka.transpose(2, 3)) / D ** 0.5
        if q_len > 1:
            t = ka.shape[2]
            i, j = torch.arange(q_len).view(-1, 1), torch.arange(t).view(1, -1)
            w = w.masked_fill(j > (t - q_len + i), float("-inf"))
        ref = torch.matmul(torch.softmax(w, dim=-1), va)
        assert out_close(out.float().cpu(), ref)
        del order[start:start + k]
        idx = torch.tensor(order)
        assert torch.equal(kk.cpu(), k_src[:, :, idx]) and torch.equal(vv.cpu(), v_src[:, :, idx])
    m = bank.slot_of_pos.cpu().numpy()
    for h in range(H):
        assert np.array_equal(np.sort(m[0, h]), np.arange(bank.cap))


def test_integration_md_ctypes_stub_binds_and_runs_verbatim():
    """The python block of INTEGRATION.md §3 is executed AS WRITTEN (its own ctypes structs, its own argument order) against
    the built library and checked against the CPU oracle, so the documented binding cannot rot."""
    from oracle import easykv_oracle as O
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    code = code.replace('"easykv_amd/csrc/libeasykv_hip.so"', repr(os.path.join(ROOT, "easykv_amd", "csrc", "libeasykv_hip.so")))
    L, Hq, H, D, P, budget = 2, 4, 4, 64, 6, 40      # budget >= 30: roco's feasible set stays clear of the 1e9 sentinels (tie-free)
    cap = 64
    g = torch.Generator().manual_seed(3)
    dev = "cuda"
    K = torch.zeros(L, H, cap, D, dtype=torch.float16, device=dev)
    V = torch.zeros_like(K)
    slot = torch.empty(L, H, cap, dtype=torch.int32, device=dev)
    S, Q, Cn = (torch.zeros(L, H, cap, dtype=torch.float32, device=dev) for _ in range(3))
    k_all, v_all, q_all = (torch.randn(L, H, P + 60, D, generator=g).half() for _ in range(3))
    K[:, :, :P], V[:, :, :P] = k_all[:, :, :P].to(dev), v_all[:, :, :P].to(dev)     # prompt rows: identity layout
    states = []
    for l in range(L):
        st = O.LayerState(k=k_all[l:l + 1, :, :P].float(), v=v_all[l:l + 1, :, :P].float())
        st.s, st.q, st.c = O.init_state_decoding((H,), budget)
        states.append(st)
    ns = dict(K=K, V=V, slot=slot, S=S, Q=Q, Cn=Cn, L=L, Hq=Hq, H=H, D=D, cap=cap, budget=budget, P=P)
    pre, per_token = code.split("# per decode token and layer l")
    exec(pre, ns)                                        # structs, bank, reset, state_init
    torch.cuda.synchronize()
    n_ev = 0
    for t in range(50):
        T_prev = P + min(t, budget)
        for l in range(L):
            q = q_all[l:l + 1, :, P + t:P + t + 1].to(dev).contiguous()
            kn = k_all[l:l + 1, :, P + t:P + t + 1].to(dev).contiguous()
            vn = v_all[l:l + 1, :, P + t:P + t + 1].to(dev).contiguous()
            out = torch.empty(1, Hq, 1, D, dtype=torch.float16, device=dev)
            ev = torch.full((1, H, 1), -1, dtype=torch.int32, device=dev)
            ns.update(l=l, T_prev=T_prev, max_T_so_far=min(P + t + 1, P + budget + 1), q=q, k_new=kn, v_new=vn, out=out, evict_ids=ev)
            exec("# per decode token and layer l" + per_token, ns)
            evict = T_prev + 1 - P > budget
            o_ref, ids_ref = O.layer_step(states[l], q.float().cpu(), kn.float().cpu(), vn.float().cpu(),
                                          O.StepPlan(policy="roco", phase="decode", evict=evict, score_off=P, budget=budget))
            assert float((out.float().cpu() - o_ref).abs().max()) < 1e-3
            if evict:
                assert torch.equal(ev[0].cpu().long(), ids_ref + P), (t, l)
                n_ev += 1
    assert n_ev == 2 * 10
