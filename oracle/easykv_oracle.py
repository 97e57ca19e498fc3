"""CPU oracle for the budgeted-KV-cache attention path of DRSY/EasyKV.

TEST INFRASTRUCTURE ONLY.  Nothing under ``easykv_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / the timed CPU baseline.

It is a restatement (not a copy) of the reference's algorithm in plain PyTorch
CPU ops, organised per policy instead of per mode.  The same torch primitives the
reference relies on (``softmax``, ``topk``, ``argmin``) are used on purpose so
that tie and NaN behaviour is inherited rather than re-invented.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` in the build container, drives it with a duck-typed
attention-only model and stores inputs + the reference's own eviction ids /
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` replays those
fixtures through this file (bit-identical ids required) and also checks the
README's structural known answers.

All ``file:line`` citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import math
import statistics
from dataclasses import dataclass, field
from typing import List, Optional

import torch

POLICIES = ("roco", "h2o_head", "tova", "recency", "random", "full")
# decode-phase recent window ratio is hard-wired (easykv/easykv.py:308, :709)
DECODE_RECENT_RATIO = 0.3
# "last 10 slots are never roco candidates" (easykv/easykv.py:321, :472, :721)
ROCO_TAIL = 10
ROCO_BIG = 1e9

# Test hook: called as hook(select_fn, policy, s, q, c, args, ids) after every selection; the golden
# generator uses it to measure how stable each decision is under relative perturbations.
SELECT_HOOK = None


# --------------------------------------------------------------------------
# a1 / a3 / a5: attention core, RoPE-on-read, GQA fold
# --------------------------------------------------------------------------
def rotate_half(x):
    """easykv/llama_patch.py:13-17."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rope_tables(seq_len: int, dim: int, base: float = 10000.0, dtype=torch.float32):
    """cos/sin tables ``[seq_len, dim]`` as HF's LlamaRotaryEmbedding(x, seq_len=) returned
    them in transformers 4.36 (third-party, not in /root/reference; call sites
    easykv/llama_patch.py:188-189, :318-319): ``emb = cat(freqs, freqs)``."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    t = torch.arange(seq_len, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def apply_rope(x, cos, sin, positions):
    """``x*cos[pos] + rotate_half(x)*sin[pos]`` (easykv/llama_patch.py:74-98).
    x ``[1,H,n,D]``, positions ``[n]``."""
    c = cos[positions].unsqueeze(0).unsqueeze(0)
    s = sin[positions].unsqueeze(0).unsqueeze(0)
    return x * c + rotate_half(x) * s


def attention_core(q, k, v, mask=None):
    """easykv/llama_patch.py:198-222 (mistral_patch.py:144-169).
    q ``[1,Hq,n,D]``; k, v ``[1,H,T,D]``; mask additive ``[1,1,n,T]`` or None.
    Returns (o ``[1,Hq,n,D]``, p ``[1,Hq,n,T]``)."""
    hq, h = q.shape[1], k.shape[1]
    rep = hq // h
    if rep > 1:  # repeat_kv, llama_patch.py:19-29
        k = k[:, :, None].expand(1, h, rep, k.shape[2], k.shape[3]).reshape(1, hq, k.shape[2], k.shape[3])
        v = v[:, :, None].expand(1, h, rep, v.shape[2], v.shape[3]).reshape(1, hq, v.shape[2], v.shape[3])
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(q.shape[-1])
    if mask is not None:
        w = w + mask
    p = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(p, v)
    return o, p


def attention_core_stream(q, k_unrot, v, cos, sin, mask=None):
    """Streaming variant, easykv/llama_patch.py:310-327: keys are cached un-rotated and
    rotated at read time with slot-index positions 0..T-1; the n queries take positions
    T-n..T-1."""
    t, n = k_unrot.shape[2], q.shape[2]
    qpos = torch.arange(t - n, t)
    kpos = torch.arange(0, t)
    return attention_core(apply_rope(q, cos, sin, qpos), apply_rope(k_unrot, cos, sin, kpos), v, mask)


def causal_chunk_mask(n: int, t: int, dtype=torch.float32):
    """Additive ``[1,1,n,t]`` mask for n new queries over t slots (the last n slots are the
    chunk itself, causal inside it).  HF builds it from the 2-D ones mask the driver passes
    (easykv/easykv.py:430); masked entries are ``finfo.min``."""
    if n == 1:
        return torch.zeros(1, 1, 1, t, dtype=dtype)
    m = torch.zeros(n, t, dtype=dtype)
    i = torch.arange(n).unsqueeze(1)
    j = torch.arange(t).unsqueeze(0)
    m[j > (t - n + i)] = torch.finfo(dtype).min
    return m.view(1, 1, n, t)


def gqa_fold(p, n_kv_heads: int, rep: int):
    """easykv/easykv.py:188-196 (inline :271-276, :683-688): mean over the query heads of a
    KV group, in p's dtype.  ``[1,Hq,n,T] -> [1,H,n,T]``."""
    return p.reshape(p.shape[0], n_kv_heads, rep, p.shape[2], p.shape[3]).mean(dim=2)


# --------------------------------------------------------------------------
# a13: budget geometry
# --------------------------------------------------------------------------
def _largest_idx(length: int, budget_p: int, stride: int) -> int:
    for idx in range(budget_p, -1, -1):
        if (length - idx) % stride == 0:
            return idx
    raise ValueError("no idx")


def geometry_encoding(length: int, budget, stride: int):
    """easykv/easykv.py:385-392 -> (budget', idx, r_idx) with the LARGEST r_idx."""
    budget_p = int(length * budget) + stride if isinstance(budget, float) else budget + stride
    idx = _largest_idx(length, budget_p, stride)
    r_idx = None
    for r in range(idx - 1, -1, -1):
        if (idx - r) % stride == 0:
            r_idx = r
            break
    return budget_p, idx, r_idx


def geometry_auto(length: int, budget: int, stride: int):
    """easykv/easykv.py:544-552 -> (budget', idx, r_idx) with the SMALLEST r_idx >= 1."""
    budget_p = budget + stride
    if budget_p >= length:
        budget_p -= stride
    idx = _largest_idx(length, budget_p, stride)
    r_idx = None
    for r in range(1, idx):
        if (idx - r) % stride == 0:
            r_idx = r
            break
    return budget_p, idx, r_idx


def geometry_ppl(length: int, budget, stride: int):
    """easykv/easykv.py:773-780."""
    budget_p = int(length * budget) + stride if isinstance(budget, float) else budget + stride
    idx = _largest_idx(length, budget_p, stride)
    r_idx = None
    for r in range(1, idx):
        if (idx - r) % stride == 0:
            r_idx = r
            break
    return budget_p, idx, r_idx


# --------------------------------------------------------------------------
# a6: score state (works on any leading shape [..., W])
# --------------------------------------------------------------------------
def init_state_decoding(lead, budget: int, device="cpu"):
    """easykv/easykv.py:242-245: zeros, count[j] = budget - j, W = budget+1."""
    w = budget + 1
    s = torch.zeros(*lead, w, device=device)
    q = torch.zeros(*lead, w, device=device)
    c = (torch.arange(w - 1, -1, -1, device=device, dtype=torch.float32)).expand(*lead, w).clone()
    return s, q, c


def init_state_prefill(lead, idx: int, stride: int, keep_attention: bool, prefix_maps=None, device="cpu"):
    """easykv/easykv.py:405 (+ h2o_head_score :173-186) and :412-416.  ``prefix_maps`` is the
    GQA-folded prefix probability tensor ``[*lead, r, r]`` when keep_attention."""
    w = idx + stride
    s = torch.zeros(*lead, w, device=device)
    q = torch.zeros(*lead, w, device=device)
    if keep_attention:
        r = prefix_maps.shape[-1]
        s[..., :r] = prefix_maps.sum(dim=-2)
        q[..., :r] = (prefix_maps ** 2).sum(dim=-2)
        c = torch.arange(w, 0, -1, device=device, dtype=torch.float32) - float(stride)
    else:
        c = torch.cat((torch.full((idx,), float(stride)), torch.arange(stride, 0, -1, dtype=torch.float32))).to(device) - float(stride)
    return s, q, c.expand(*lead, w).clone()


# --------------------------------------------------------------------------
# a7: accumulation
# --------------------------------------------------------------------------
def accumulate_row(policy: str, s, q, a):
    """Decode accumulate, easykv/easykv.py:287-300 (auto :693-707). ``a [..., g]`` is the folded
    probability row restricted to the scored region."""
    g = a.shape[-1]
    if policy == "h2o_head":
        s[..., :g] += a
    elif policy == "roco":
        s[..., :g] += a
        q[..., :g] += a ** 2
    elif policy == "tova":
        s[..., :g] = a


def accumulate_chunk(policy: str, s, q, pbar, tova_head_mean: bool):
    """Prefill accumulate, easykv/easykv.py:443-457 (auto :604-618, ppl :834-848).
    ``pbar [..., H, n, T]`` folded probabilities of this chunk.  ``tova_head_mean``: encoding /
    ppl share one head-averaged last-query row across heads (:456, :847); auto mode keeps it
    per head (:617)."""
    t = pbar.shape[-1]
    if policy == "h2o_head":
        s[..., :t] += pbar.sum(dim=-2)
    elif policy == "roco":
        s[..., :t] += pbar.sum(dim=-2)
        q[..., :t] += (pbar ** 2).sum(dim=-2)
    elif policy == "tova":
        last = pbar[..., -1, :]
        if tova_head_mean:
            last = last.mean(dim=-2, keepdim=True).expand_as(last)
        s[..., :t] = last


# --------------------------------------------------------------------------
# a9 / a10: victim selection
# --------------------------------------------------------------------------
def roco_std(s, q, c, sink: int = 0):
    std = torch.sqrt(q / c - (s / c) ** 2)
    std[..., -ROCO_TAIL:] = ROCO_BIG
    if sink:
        std[..., :sink] = ROCO_BIG
    return std


def select_decode(policy: str, s, q, c, budget: int):
    """easykv/easykv.py:310-337 (auto :711-740).  Returns ids ``[...]`` (one per row)."""
    ids = _select_decode(policy, s, q, c, budget)
    if SELECT_HOOK is not None:
        SELECT_HOOK(_select_decode, policy, s, q, c, (budget,), ids.unsqueeze(-1))
    return ids


def _select_decode(policy: str, s, q, c, budget: int):
    rw = int(budget * DECODE_RECENT_RATIO)
    if policy == "h2o_head":
        return torch.argmin(s[..., :-rw], dim=-1)
    if policy == "tova":
        return torch.argmin(s, dim=-1)
    if policy == "roco":
        std = roco_std(s, q, c)
        _, feas = torch.topk(std, largest=False, k=budget - rw, dim=-1)
        am = torch.argmin(s.gather(-1, feas) / c.gather(-1, feas), dim=-1, keepdim=True)
        return feas.gather(-1, am).squeeze(-1)
    raise ValueError(policy)


def select_prefill(policy: str, s, q, c, budget_p: int, recent: int, sink: int, stride: int):
    """easykv/easykv.py:462-490 (auto :623-651, ppl :853-882).  Returns ids ``[..., stride]``."""
    ids = _select_prefill(policy, s, q, c, budget_p, recent, sink, stride)
    if SELECT_HOOK is not None:
        SELECT_HOOK(_select_prefill, policy, s, q, c, (budget_p, recent, sink, stride), ids)
    return ids


def _select_prefill(policy: str, s, q, c, budget_p: int, recent: int, sink: int, stride: int):
    if policy in ("h2o_head", "tova"):
        return torch.topk(s[..., sink:-recent], dim=-1, k=stride, largest=False)[1] + sink
    if policy == "roco":
        std = roco_std(s, q, c, sink)
        _, feas = torch.topk(std, largest=False, k=max(budget_p - recent - sink, stride), dim=-1)
        am = torch.topk(s.gather(-1, feas) / c.gather(-1, feas), dim=-1, largest=False, k=stride)[1]
        return feas.gather(-1, am)
    raise ValueError(policy)


# --------------------------------------------------------------------------
# a11 / a12: compaction
# --------------------------------------------------------------------------
def _keep_mask(width: int, ids):
    """ids ``[..., k]`` -> bool ``[..., width]`` that is False at the evicted columns."""
    keep = torch.ones(*ids.shape[:-1], width, dtype=torch.bool, device=ids.device)
    return keep.scatter(-1, ids, torch.zeros_like(ids, dtype=torch.bool))


def drop_columns(x, ids, tail):
    """Order-preserving delete of ``ids`` from every row of ``x [..., W]`` then append ``tail [k]``
    (easykv/easykv.py:315-318, :328-333, :465-469, :478-483)."""
    k = ids.shape[-1]
    keep = _keep_mask(x.shape[-1], ids)
    body = x[keep].view(*x.shape[:-1], x.shape[-1] - k)
    return torch.cat((body, tail.to(x.dtype).expand(*x.shape[:-1], k)), dim=-1)


def drop_kv_slots(kv, ids):
    """Per-head order-preserving delete (easykv/easykv.py:56-82).  kv ``[1,H,T,D]``; ids ``[H,k]``."""
    _, h, t, d = kv.shape
    keep = _keep_mask(t, ids)
    return kv[0][keep].view(1, h, t - ids.shape[-1], d)


def drop_kv_range(kv, start: int, end: int):
    """Same contiguous range for every head (easykv/easykv.py:105-112)."""
    return torch.cat((kv[:, :, :start], kv[:, :, end:]), dim=2)


# --------------------------------------------------------------------------
# Per-layer step: what ONE fused HIP launch must reproduce for one layer.
# --------------------------------------------------------------------------
@dataclass
class StepPlan:
    """Everything the driver decides for one model forward (same for every layer)."""
    policy: str = "roco"
    phase: str = "decode"          # "decode" | "prefill"
    accumulate: bool = True
    evict: bool = False
    score_off: int = 0             # P: first logical slot covered by the state rows
    budget: int = 0                # decode: budget; prefill: budget'
    recent: int = 0                # prefill only
    sink: int = 0                  # prefill only
    stride: int = 1
    tova_head_mean: bool = False
    range_start: int = -1          # recency / random: evict [range_start, range_start+stride)
    streaming: bool = False


@dataclass
class LayerState:
    """One layer's retained cache (birth order) and score rows."""
    k: torch.Tensor                # [1,H,T,D]
    v: torch.Tensor
    s: Optional[torch.Tensor] = None   # [H,W]
    q: Optional[torch.Tensor] = None
    c: Optional[torch.Tensor] = None


def layer_step(st: LayerState, qn, kn, vn, plan: StepPlan, cos=None, sin=None):
    """One layer, one forward: append -> attention -> fold -> accumulate -> select -> compact.
    qn ``[1,Hq,n,D]``, kn/vn ``[1,H,n,D]``.  Returns (o ``[1,Hq,n,D]``, ids or None).
    Probe 7 of SURVEY.md verified that doing this layer by layer equals the reference's
    all-layers-at-once order."""
    h = kn.shape[1]
    rep = qn.shape[1] // h
    n = qn.shape[2]
    st.k = torch.cat((st.k, kn), dim=2)
    st.v = torch.cat((st.v, vn), dim=2)
    t = st.k.shape[2]
    mask = causal_chunk_mask(n, t, qn.dtype)
    if plan.streaming:
        o, p = attention_core_stream(qn, st.k, st.v, cos, sin, mask)
    else:
        o, p = attention_core(qn, st.k, st.v, mask)
    ids = None
    if plan.policy in ("roco", "h2o_head", "tova"):
        pbar = gqa_fold(p, h, rep)[0]          # [H,n,T]
        if plan.accumulate:
            if plan.phase == "decode":
                accumulate_row(plan.policy, st.s, st.q, pbar[:, 0, plan.score_off:])
            else:
                accumulate_chunk(plan.policy, st.s, st.q, pbar, plan.tova_head_mean)
        if plan.evict:
            if plan.phase == "decode":
                st.c += 1.0
                ids = select_decode(plan.policy, st.s, st.q, st.c, plan.budget).unsqueeze(-1)
                tail_c = torch.zeros(1)
            else:
                st.c += float(plan.stride)
                ids = select_prefill(plan.policy, st.s, st.q, st.c, plan.budget, plan.recent, plan.sink, plan.stride)
                tail_c = -torch.arange(plan.stride, dtype=torch.float32)
            k = ids.shape[-1]
            st.k = drop_kv_slots(st.k, ids + plan.score_off)
            st.v = drop_kv_slots(st.v, ids + plan.score_off)
            st.s = drop_columns(st.s, ids, torch.zeros(k))
            if plan.policy == "roco":
                st.q = drop_columns(st.q, ids, torch.zeros(k))
            if plan.policy == "roco" or (plan.policy == "h2o_head" and plan.phase == "prefill"):
                st.c = drop_columns(st.c, ids, tail_c)
    elif plan.evict and plan.range_start >= 0:
        st.k = drop_kv_range(st.k, plan.range_start, plan.range_start + plan.stride)
        st.v = drop_kv_range(st.v, plan.range_start, plan.range_start + plan.stride)
    return o, ids


# --------------------------------------------------------------------------
# Driver (reference protocol): restates easykv/easykv.py:199-901 for a duck-typed model.
# --------------------------------------------------------------------------
@dataclass
class Trace:
    """What the parity tests compare."""
    evictions: List[dict] = field(default_factory=list)   # {"step", "kind", "ids"|"range"}
    cache_len: int = 0
    n_forwards: int = 0
    result: object = None
    report: str = ""


def _fold_all(attns, n_layers, h, rep):
    return torch.stack([gqa_fold(attns[l], h, rep)[0] for l in range(n_layers)])  # [L,H,n,T]


def _greedy(logits_last):
    """The sampler (easykv/easykv.py:115-134) is out of scope; the fixtures use one-hot
    logits so multinomial sampling is deterministic and equals argmax."""
    return torch.argmax(logits_last, dim=-1, keepdim=True)


def _kv_apply(past, fn):
    return tuple((fn(k), fn(v)) for (k, v) in past)


def _kv_drop_per_head(past, ids_lhk, off=0):
    return tuple((drop_kv_slots(k, ids_lhk[l] + off), drop_kv_slots(v, ids_lhk[l] + off)) for l, (k, v) in enumerate(past))


def generate(self, input_ids, generation_config, kv_mode="encoding", stride=1):
    """Restatement of ``generate`` (easykv/easykv.py:199-901) against the duck-typed model
    contract of SURVEY.md §8(b).  Returns a :class:`Trace`."""
    cfg = generation_config
    max_new = cfg.get("max_new_tokens", 1024)
    budget = cfg.get("budget", 0.5)
    policy = cfg.get("kv_policy", "recency")
    sink = cfg.get("temp_length", 4)
    recent_ratio = cfg.get("recent_ratio", 0.1)
    keep_attention = cfg.get("keep_attention", False)
    eos = cfg.get("eos_token_ids", [self.tokenizer.eos_token_id])
    n_layers = self.config.num_hidden_layers
    hq = self.config.num_attention_heads
    h = getattr(self.config, "num_key_value_heads", hq)
    rep = hq // h
    dev = self.device
    tr = Trace()
    length = input_ids.shape[-1]

    if kv_mode == "auto":                                   # :220-227
        assert type(budget) == int
        if budget > length:
            kv_mode, budget = "decoding", budget - length
        else:
            kv_mode = "encoding_decoding"

    def forward(ids, past, pos, want_attn):
        tr.n_forwards += 1
        t_prev = 0 if past is None else past[0][0].shape[2]
        return self(input_ids=ids, past_key_values=past,
                    attention_mask=torch.ones(1, ids.shape[1] + t_prev, dtype=torch.long, device=dev),
                    position_ids=torch.as_tensor(pos, dtype=torch.long, device=dev).view(1, -1),
                    use_cache=True, output_attentions=want_attn)

    scored = policy in ("roco", "h2o_head", "tova")

    # ---- single-token decode with eviction (decoding mode and the tail of auto mode) ----
    def decode_loop(past, logits_last, cur_pos, s, q, c, score_off, budget_d, whole_cache):
        positions = []                                       # cache_positions, :241
        out_ids = []
        n = 0
        while n < max_new:                                   # :257 / :670
            tok = _greedy(logits_last)
            out_ids.append(int(tok[0, 0]))
            n += 1
            if out_ids[-1] in eos:
                break
            out = forward(tok, past, [cur_pos], True)
            past = out.past_key_values
            logits_last = out.logits[:, -1, :]
            positions.append(cur_pos)
            if scored:
                pbar = _fold_all(out.attentions, n_layers, h, rep)          # [L,H,1,T]
                accumulate_row(policy, s, q, pbar[:, :, 0, score_off:])     # :287-300 / :693-707
            t_now = past[0][0].shape[2]
            evict = whole_cache or ((t_now - score_off) > budget_d and policy != "full")   # :303 / always :708
            if whole_cache and policy == "full":
                evict = False
            if evict:
                c += 1.0                                                     # :304 / :708
                if scored:
                    ids = select_decode(policy, s, q, c, budget_d)           # [L,H]
                    tr.evictions.append({"step": tr.n_forwards, "kind": "per_head", "ids": (ids + score_off).unsqueeze(-1).clone()})
                    past = _kv_drop_per_head(past, ids.unsqueeze(-1), score_off)
                    z = torch.zeros(1, device=dev)
                    s = drop_columns(s, ids.unsqueeze(-1), z)
                    if policy == "roco":
                        q = drop_columns(q, ids.unsqueeze(-1), z)
                        c = drop_columns(c, ids.unsqueeze(-1), z)
                elif policy in ("recency", "random"):
                    if whole_cache:                                          # :741-747
                        if policy == "random":
                            raise UnboundLocalError("auto+random: positions_tensor undefined (easykv/easykv.py:744)")
                        start = sink
                    else:                                                    # :343-362
                        pt = torch.tensor(positions, device=dev).float() / float(cur_pos)
                        scores = (1.0 - pt) if policy == "recency" else torch.rand(*pt.shape).to(dev)
                        e = int(torch.topk(scores, k=1, dim=-1)[1][0])
                        positions.pop(e)
                        start = score_off + e
                    tr.evictions.append({"step": tr.n_forwards, "kind": "range", "range": (start, start + 1)})
                    past = _kv_apply(past, lambda x: drop_kv_range(x, start, start + 1))
            cur_pos += 1
        return past, out_ids

    # ---- strided prefill with eviction (encoding, auto, ppl) ----
    def prefill_loop(budget_p, idx, r_idx, tova_head_mean, keep_logits=False):
        recent = int(budget_p * recent_ratio)                                # :394
        out = self(input_ids=input_ids[:, :r_idx], use_cache=True, output_attentions=keep_attention)   # :396
        tr.n_forwards += 1
        past, logits_last = out.past_key_values, out.logits[:, -1, :]
        maps = _fold_all(out.attentions, n_layers, h, rep) if keep_attention else None
        s, q, c = init_state_prefill((n_layers, h), idx, stride, keep_attention, maps, dev)
        cur_pos = past[0][0].shape[2]
        all_logits, all_ids = [], []
        for tok_i in range(r_idx, length, stride):                           # :426
            out = forward(input_ids[:, tok_i:tok_i + stride], past, list(range(cur_pos, cur_pos + stride)), True)
            past, logits_last = out.past_key_values, out.logits[:, -1, :]
            if keep_logits:
                all_logits.append(out.logits[0])
                all_ids.append(input_ids[0, tok_i:tok_i + stride])
            t_now = past[0][0].shape[2]
            if scored and (t_now > idx or keep_attention):                   # :443
                accumulate_chunk(policy, s, q, _fold_all(out.attentions, n_layers, h, rep), tova_head_mean)
            if policy != "full" and t_now > idx:                             # :459
                c += float(stride)
                if scored:
                    ids = select_prefill(policy, s, q, c, budget_p, recent, sink, stride)   # [L,H,s]
                    tr.evictions.append({"step": tr.n_forwards, "kind": "per_head", "ids": ids.clone()})
                    past = _kv_drop_per_head(past, ids)
                    s = drop_columns(s, ids, torch.zeros(stride, device=dev))
                    if policy == "roco":
                        q = drop_columns(q, ids, torch.zeros(stride, device=dev))
                    if policy in ("roco", "h2o_head"):
                        c = drop_columns(c, ids, -torch.arange(stride, dtype=torch.float32, device=dev))
                elif policy in ("recency", "random"):
                    if policy == "recency":
                        start = sink                                         # :491-493
                    else:                                                    # :494-499
                        sc = torch.rand(s.shape[-1]).to(dev)
                        sc[-stride:] = -1e9
                        start = int(torch.topk(sc, k=1, dim=-1)[1][0])
                    tr.evictions.append({"step": tr.n_forwards, "kind": "range", "range": (start, start + stride)})
                    past = _kv_apply(past, lambda x: drop_kv_range(x, start, start + stride))
            cur_pos += stride
        return past, logits_last, s, q, c, all_logits, all_ids

    if kv_mode == "decoding":                                                # :228-366
        out = self(input_ids=input_ids, use_cache=True)
        tr.n_forwards += 1
        past, logits_last = out.past_key_values, out.logits[:, -1, :]
        s, q, c = init_state_decoding((n_layers, h), budget, dev)
        past, out_ids = decode_loop(past, logits_last, past[0][0].shape[2], s, q, c, length, budget, False)
        kept = past[0][0].shape[2] - length
        tr.report = f"KV cache budget ratio: {kept / len(out_ids) * 100:.2f}%({kept}/{len(out_ids)})"
        tr.result = out_ids

    elif kv_mode == "encoding":                                              # :367-529
        full = (type(budget) == float and budget >= 1.0) or (type(budget) == int and budget >= length)
        if full:
            out = self(input_ids=input_ids, use_cache=True)
            tr.n_forwards += 1
            past, logits_last = out.past_key_values, out.logits[:, -1, :]
        else:
            budget_p, idx, r_idx = geometry_encoding(length, budget, stride)
            past, logits_last, *_ = prefill_loop(budget_p, idx, r_idx, True)
        kept = past[0][0].shape[2]
        tr.report = f"KV cache budget ratio: {kept / length * 100:.2f}%({kept}/{length})"
        cur_pos, out_ids, n = length, [], 0
        while n < max_new:                                                   # :508-526 plain decode
            tok = _greedy(logits_last)
            out_ids.append(int(tok[0, 0]))
            n += 1
            if out_ids[-1] in eos:
                break
            out = forward(tok, past, [cur_pos], False)
            past, logits_last = out.past_key_values, out.logits[:, -1, :]
            cur_pos += 1
        tr.result = out_ids

    elif kv_mode == "encoding_decoding":                                     # :530-753
        assert type(budget) == int and budget <= length
        assert policy in ["random", "recency", "tova", "roco"]
        budget_p, idx, r_idx = geometry_auto(length, budget, stride)
        past, logits_last, s, q, c, *_ = prefill_loop(budget_p, idx, r_idx, False)
        kept = past[0][0].shape[2]
        s, q, c = s[..., :-(stride - 1)], q[..., :-(stride - 1)], c[..., :-(stride - 1)]   # :666-668
        assert s.shape[-1] == kept + 1                                       # :669
        past, out_ids = decode_loop(past, logits_last, length, s.clone(), q.clone(), c.clone(), 0, budget_p, True)
        size = past[0][0].shape[2]
        tr.report = f"KV Cache Budget ratio {size / (length + len(out_ids)) * 100:.2f}%[{size}/({length}+{len(out_ids)})]"
        tr.result = out_ids

    elif kv_mode == "ppl":                                                   # :754-901
        ce = torch.nn.CrossEntropyLoss(reduction="none")
        if budget >= 1.0:
            out = self(input_ids=input_ids, use_cache=False)
            tr.n_forwards += 1
            lp = ce(out.logits[0, :-1], input_ids.clone()[0, 1:]).cpu().numpy().tolist()
            tr.result = math.exp(statistics.mean(lp))
            tr.cache_len = length
            return tr
        budget_p, idx, r_idx = geometry_ppl(length, budget, stride)
        past, _, _, _, _, all_logits, all_ids = prefill_loop(budget_p, idx, r_idx, True, keep_logits=True)
        kept = past[0][0].shape[2]
        tr.report = f"KV cache budget ratio: {kept / length * 100:.2f}%({kept}/{length})"
        ids_cat, log_cat = torch.cat(all_ids), torch.cat(all_logits, dim=0)
        assert ids_cat.shape[0] == log_cat.shape[0]
        lp = ce(log_cat[:-1], ids_cat[1:]).cpu().numpy().tolist()
        tr.result = math.exp(statistics.mean(lp))
    else:
        raise ValueError(kv_mode)

    tr.cache_len = past[0][0].shape[2]
    tr.final_k = [k for (k, _) in past]
    return tr
