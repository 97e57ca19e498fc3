"""Host-side mirror of the reference's public surface for the budgeted-KV path.

    from easykv_amd import enable_fixed_kv
    enable_fixed_kv(model, tokenizer, mode='auto', stride=8)
    text = model.easykv_generate(input_ids=ids, generation_config=dict(budget=2048, kv_policy='roco'))

Same names, argument meaning, generation_config keys/defaults, kv_policy strings, return types and
printed lines as the reference (easykv/easykv.py:199-210, :903-908).  What differs is WHERE the work
happens: the reference asks the model for every layer's probability matrix and scores / selects /
compacts in Python (easykv/easykv.py:264-362); here the driver only decides the per-forward
:class:`StepPlan` and every layer's attention call runs the fused HIP step on a device-resident
:class:`KVBank` — no attention map, no host sync, no K/V copy per step.

Model contract (what ``self`` must provide; the reference's is SURVEY.md §8b):
  * ``self.config.{num_hidden_layers, num_attention_heads[, num_key_value_heads][, head_dim | hidden_size]}``,
    ``self.device`` (a GPU), ``self.tokenizer.{eos_token_id, decode}``;
  * ``self(input_ids=, past_key_values=<BudgetedKVCache>, position_ids=, use_cache=True)`` returning an object
    with ``.logits [1, n, V]``; inside, every attention layer calls
    ``past_key_values.attend(layer_idx, q [1,Hq,n,D], k [1,H,n,D], v [1,H,n,D]) -> [1,Hq,n,D]`` with keys already
    rotated by their TRUE positions (un-rotated when ``generation_config['streaming']``: the kernel then rotates at
    read time by slot index, easykv/llama_patch.py:310-327).
  ``easykv_amd.hf`` adapts HF transformers >= 5 Llama/Mistral models to this contract.
"""
from __future__ import annotations

import contextlib
import contextvars
import functools
import math
import statistics
import time
from typing import List, Optional

import torch

from .engine import KVBank, StepPlan

KNOWN_POLICIES = ("roco", "h2o_head", "tova", "recency", "random", "full")
SCORED = ("roco", "h2o_head", "tova")
PREFIX_BLOCK = 64   # queries per launch when the prefix must be scored (keep_attention)


# ------------------------------------------------------------------------------------------------
# budget geometry (easykv/easykv.py:385-392, :544-552, :773-780)
# ------------------------------------------------------------------------------------------------
def _idx_for(length: int, budget_p: int, stride: int) -> int:
    return next(i for i in range(budget_p, -1, -1) if (length - i) % stride == 0)


def geometry(mode: str, length: int, budget, stride: int):
    """-> (budget', idx, r_idx).  ``idx`` = retained slots after the strided prefill, ``r_idx`` = dense prefix."""
    if isinstance(budget, float):
        budget_p = int(length * budget) + stride
    else:
        budget_p = budget + stride
        if mode == "auto" and budget_p >= length:
            budget_p -= stride
    idx = _idx_for(length, budget_p, stride)
    if mode == "encoding":      # largest r_idx < idx on the stride grid (:391-392)
        r_idx = next((r for r in range(idx - 1, -1, -1) if (idx - r) % stride == 0), None)
    else:                       # auto / ppl: smallest r_idx >= 1 (:551-552, :779-780)
        r_idx = next((r for r in range(1, idx) if (idx - r) % stride == 0), None)
    return budget_p, idx, r_idx


# ------------------------------------------------------------------------------------------------
# the cache object handed to the model
# ------------------------------------------------------------------------------------------------
# The cache of the model forward in flight, for attention seams that are not handed ``past_key_values`` (easykv_amd.hf's
# AttentionInterface function).  A context variable, set around every forward by :func:`generate` and reset afterwards: two
# models, nested or interleaved generates and threads cannot see each other's cache, and a forward outside easykv_generate()
# finds nothing (the seam then raises instead of appending into a stale bank).
_ACTIVE: contextvars.ContextVar = contextvars.ContextVar("easykv_amd_active_cache", default=None)


def active_cache():
    """The :class:`BudgetedKVCache` of the forward in flight in this context, or None."""
    return _ACTIVE.get()


class BudgetedKVCache:
    """Device-resident budgeted cache of one sequence.  The driver sets :attr:`plan` before each model
    forward; every attention layer then calls :meth:`attend`.  It also duck-types the two ``Cache`` methods HF
    transformers >= 5 calls on ``past_key_values`` (``update`` hands the new rows straight back: the bank appends them
    inside :meth:`attend`).

    ``layer_begin`` / ``layer_count``: the contiguous block of the model's layers THIS process owns (layer sharding,
    SURVEY.md §8e: the reference spreads layers over GPUs with ``device_map='auto'``, test_passkey.py:25-35; here one process
    per GPU owns a block, easykv_amd/dist.py).  The bank holds only those layers; :meth:`attend` takes GLOBAL layer indices."""

    def __init__(self, n_layers, n_q_heads, n_kv_heads, head_dim, cap, device, streaming=False, rope=None,
                 record=False, layer_begin=0, layer_count=None, rope_base=10000.0):
        self.layer_begin = layer_begin
        self.layer_count = n_layers - layer_begin if layer_count is None else layer_count
        if not (0 <= self.layer_begin and self.layer_count >= 1 and self.layer_begin + self.layer_count <= n_layers):
            raise ValueError(f"layer block [{layer_begin}, {layer_begin}+{layer_count}) outside the model's {n_layers} layers")
        self.n_model_layers = n_layers
        self.bank = KVBank(self.layer_count, n_q_heads, n_kv_heads, head_dim, cap, device=device)
        self.plan = StepPlan(policy="full", phase="prefill", accumulate=False)
        self.streaming = streaming
        self.positions = None    # true position ids of the forward in flight (set by the driver)
        self.unrotate = None     # HF seam + streaming: (cos, sin) fp32 [>= max position, D] to take the model's RoPE off q/k
        if streaming:
            cos, sin = rope[:2] if rope is not None else rope_tables(cap, head_dim, rope_base)
            self.bank.set_rope(cos, sin)
        self.record = record
        self.evictions = []      # record=True: per forward with eviction: list over OWNED layers of int32 [H,k] (device)
        self._cur = None
        self.score_prefix = False
        self.n_attend = 0        # attend() calls of the forward in flight (checked by the driver after every forward)
        self._defer_this_forward = None
        self.defer_chunk_scorer = True     # scored chunk steps of a layer-per-call model: one scorer launch per forward (round 4)

    def owns(self, layer_idx: int) -> bool:
        return self.layer_begin <= layer_idx < self.layer_begin + self.layer_count

    def get_seq_length(self, layer_idx: Optional[int] = None) -> int:
        """Live slots (every owned layer holds the same number, as in the reference)."""
        return self.bank.n_slots[0 if layer_idx is None or not self.owns(layer_idx) else layer_idx - self.layer_begin]

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        return key_states, value_states

    def begin_forward(self, plan: StepPlan, positions=None):
        self.bank.abort_step()     # a previous forward that raised between two layers leaves a half-open deferred token step
        self.plan = plan
        self.positions = positions
        self.n_attend = 0
        self._defer_this_forward = None      # decided by the forward's first attend() (all owned layers still hold the same length)
        self._cur = [] if (self.record and plan.evict) else None
        if self._cur is not None:
            self.evictions.append(self._cur)

    @contextlib.contextmanager
    def active(self, plan: StepPlan, positions=None):
        """Scope of ONE model forward driven by hand (generate() does this around every forward it issues): sets the plan
        and makes this cache the one attention seams without a ``past_key_values`` argument (easykv_amd.hf) find."""
        self.begin_forward(plan, positions)
        tok = _ACTIVE.set(self)
        try:
            yield self
        finally:
            _ACTIVE.reset(tok)

    def _prefix_in_one_step(self, plan, n, layer_idx) -> bool:
        key = (plan.policy, plan.accumulate, plan.two_pass, type(self.bank).default_two_pass, self.streaming, n)
        if getattr(self, "_prefix_rule", (None, None))[0] != key:
            info = self.bank.step_info(plan, n, layer_idx, 1)
            self._prefix_rule = (key, bool(info["two_pass"] and info["wide"]))
        return self._prefix_rule[1]

    DEFER_WORKSPACE_LIMIT = 1 << 30      # bytes of logits / column sums of all layers a deferred chunk step may keep alive ...
    DEFER_WORKSPACE_FRACTION = 1 / 8     # ... and never more than this share of the device memory that is free when the rule is made

    def _defer_fits(self, plan, n) -> bool:
        """Does this scored chunk step run with the scorers of all owned layers deferred to one launch per forward?  Not when the
        immediate form is ONE launch already (the logits-in-LDS kernel, a chunk step with the scorer as its kernel tail — ekv_step_info
        ``fused``: deferral would take the step off that kernel, whose arithmetic and speed differ), and not when every layer's logits /
        column sums kept until the flush would pin too much HBM (one-pass shapes of long caches)."""
        key = (plan.policy, plan.accumulate, plan.evict, plan.two_pass, type(self.bank).default_two_pass, self.streaming, n, self.bank.n_slots[0])
        if getattr(self, "_defer_rule", (None, None))[0] != key:
            one_launch = bool(self.bank.step_info(plan, n, 0, 1)["fused"])
            limit = min(self.DEFER_WORKSPACE_LIMIT, int(torch.cuda.mem_get_info(self.bank.device)[0] * self.DEFER_WORKSPACE_FRACTION))
            self._defer_rule = (key, (not one_launch) and self.bank.deferred_workspace_bytes(plan, n) <= limit)
        return self._defer_rule[1]

    def attend(self, layer_idx: int, q, k, v):
        """One layer of one forward: append + attention + score + select + compaction, all on device.
        ``layer_idx`` is the layer's index in the MODEL; it must lie in this cache's block."""
        if not self.owns(layer_idx):
            raise ValueError(f"layer {layer_idx} is not in this rank's block [{self.layer_begin}, {self.layer_begin + self.layer_count})")
        if q.shape[0] != 1 or k.shape[0] != 1 or v.shape[0] != 1:
            raise ValueError("the budgeted-KV path is batch-size 1 (as the reference: easykv/easykv.py asserts nothing but indexes [0])")
        layer_idx -= self.layer_begin
        self.n_attend += 1
        plan = self.plan
        n = q.shape[2]
        # fp16 rows are read IN PLACE at their strides (ABI 8): HF hands over [1, H, n, D] transposed views of the projections'
        # [1, n, H * D] output — three copy kernels per layer in front of every chunk step until round 5.  Other dtypes are converted;
        # layouts the kernels cannot address are copied dense by KVBank.attend.
        q, k, v = (t if t.dtype == torch.float16 else t.to(torch.float16) for t in (q, k, v))
        # the output is written token-major ([1, n, Hq, D]) and returned as its [1, Hq, n, D] view: the transpose(1, 2) every caller
        # applies next (easykv_amd.hf, llama_patch.py:230-232) is then a dense tensor, no copy in front of o_proj
        out = torch.empty(1, n, q.shape[1], q.shape[3], dtype=torch.float16, device=q.device).transpose(1, 2) if n > 1 else None
        if self.score_prefix and n > PREFIX_BLOCK and not self._prefix_in_one_step(plan, n, layer_idx):
            # keep_attention: the dense prefix must also feed the score rows (easykv/easykv.py:173-186).  ONE launch pair per layer
            # when the LIBRARY says the step runs as the two-pass scheme on the wide-block kernel (ekv_step_info: the query blocks
            # are walked inside the launch and nothing of size r x r exists anywhere); every other dispatch (RoPE-on-read,
            # head_dim 32, odd GQA factors, EKV_NO_WIDE, a forced one-pass scheme) exports logits or one column-sum row per query
            # block to a workspace, so there the prefix is cut into query blocks here
            outs = []
            for i0 in range(0, n, PREFIX_BLOCK):
                self.bank.attend(plan, q[:, :, i0:i0 + PREFIX_BLOCK], k[:, :, i0:i0 + PREFIX_BLOCK], v[:, :, i0:i0 + PREFIX_BLOCK],
                                 layer_begin=layer_idx, out=out[:, :, i0:i0 + PREFIX_BLOCK])
            return out
        if self._defer_this_forward is None:      # one decision per forward: an immediate step advances its layer's length at once
            self._defer_this_forward = bool(n > 1 and plan.phase == "prefill" and plan.policy in ("roco", "h2o_head", "tova") and (plan.accumulate or plan.evict)
                                            and self.defer_chunk_scorer and self._defer_fits(plan, n))
        deferable_chunk = self._defer_this_forward
        if ((n == 1 and plan.phase == "decode") or deferable_chunk) and self.layer_count > 1:
            # one layer per call (a decoder stack): attention + fold of this layer now, the scorers of all owned layers in ONE
            # launch after the last layer (KVBank.flush) — off the critical path of the stack.  Decode steps, and since round 4 the
            # scored chunk steps of a strided prefill (their scorer launch — 32 workgroups, latency-bound, folds the key-range
            # partials too — was more than half of a layer's time)
            out, _ = self.bank.attend(plan, q, k, v, layer_begin=layer_idx, defer=True, out=out)
            if self.n_attend == self.layer_count:
                ids = self.bank.flush()
                if self._cur is not None and ids is not None:
                    self._cur.extend(ids[l] for l in range(self.layer_count))
            return out
        # (the evicted cache indices are only wanted when a caller records them: _record_evictions)
        out, ids = self.bank.attend(plan, q, k, v, layer_begin=layer_idx, evict_ids=None if self._cur is not None else False, out=out)
        if self._cur is not None and ids is not None:
            self._cur.append(ids[0])
        return out


def rope_tables(seq_len: int, dim: int, base: float = 10000.0):
    """fp32 cos/sin ``[seq_len, dim]`` with the HF layout ``cat(freqs, freqs)``."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    freqs = torch.outer(torch.arange(seq_len, dtype=torch.float32), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


# ------------------------------------------------------------------------------------------------
# sampler (easykv/easykv.py:115-134) — vocab-sized torch ops, not part of the KV path
# ------------------------------------------------------------------------------------------------
def logits_adapter(logits: torch.Tensor, temperature: float, top_p: float):
    prob = torch.softmax(logits / temperature, dim=-1)
    sorted_prob, order = torch.sort(prob, descending=True, dim=-1)
    keep = (torch.cumsum(sorted_prob, dim=-1) - sorted_prob) <= top_p
    sorted_prob = sorted_prob * keep
    sorted_prob = sorted_prob / sorted_prob.sum(dim=-1, keepdim=True)
    final = torch.zeros_like(prob).scatter(-1, order, sorted_prob)
    return final, torch.softmax(logits, dim=-1)


def _dims(self):
    cfg = self.config
    n_layers = cfg.num_hidden_layers
    hq = cfg.num_attention_heads
    h = getattr(cfg, "num_key_value_heads", None) or hq
    d = getattr(cfg, "head_dim", None) or (cfg.hidden_size // hq)
    return n_layers, hq, h, d


# ------------------------------------------------------------------------------------------------
# generate
# ------------------------------------------------------------------------------------------------
@torch.inference_mode()
def generate(self, input_ids, generation_config, kv_mode="encoding", stride=1, report_decoding_latency: bool = False,
             return_cache: bool = False):
    cfg = generation_config
    temperature = cfg.get("temperature", 1.0)
    top_p = cfg.get("top_p", 1.0)
    max_new_tokens = cfg.get("max_new_tokens", 1024)
    budget = cfg.get("budget", 0.5)
    policy = cfg.get("kv_policy", "recency")
    sink = cfg.get("temp_length", 4)
    recent_ratio = cfg.get("recent_ratio", 0.1)
    keep_attention = cfg.get("keep_attention", False)
    eos_token_ids = cfg.get("eos_token_ids", [self.tokenizer.eos_token_id])
    streaming = cfg.get("streaming", False)
    record = cfg.get("_record_evictions", False)      # test hook: keep the evicted ids of every forward
    # extension key: capture the steady-state forwards — the evicting decode step and, round 6, the evicting strided chunk of the
    # prefill — in a hipGraph each (GraphedForward)
    use_graph = cfg.get("hipgraph", False)
    n_layers, hq, h, d = _dims(self)
    dev = torch.device(self.device)
    if input_ids.dim() != 2 or input_ids.shape[0] != 1:
        raise ValueError(f"input_ids must be [1, S] (batch size 1, as the reference); got {tuple(input_ids.shape)}")
    length = input_ids.shape[-1]
    input_ids = input_ids.to(dev)
    scored = policy in SCORED
    evicting = policy in KNOWN_POLICIES and policy != "full"   # unknown strings evict nothing

    # Layer sharding: a model that carries ``layer_shard`` (easykv_amd.dist.LayerShard) runs only its own
    # block of layers in this process; every rank drives the same loop (the plans depend on lengths only), the bank of a
    # rank holds its layers only, the model's forward moves the stage output to the next rank, and the sampled token comes
    # from the last stage.
    shard = getattr(self, "layer_shard", None)
    if shard is not None and shard.world == 1:
        shard = None
    if shard is not None:
        if shard.world > n_layers or shard.count < 1:
            raise ValueError(f"layer sharding over {shard.world} ranks needs at least one of the model's {n_layers} layers per rank")
        if use_graph:
            # a captured forward would contain the stage hand-off (dist.send / dist.recv and, on gloo, host staging)
            raise ValueError("generation_config['hipgraph'] is not supported on a layer-sharded model (model.layer_shard)")
    l_begin, l_count = (shard.begin, shard.count) if shard is not None else (0, n_layers)

    def policy_draw(n, mask_tail=0):
        """kv_policy='random': the reference's own draw — argmax of torch.rand on the global CPU generator over the row
        (easykv/easykv.py:354-356; the chunk's own columns excluded in prefill, :494-497).  Layer-sharded: every rank draws
        (generators seeded alike stay in step) and rank 0's value is the one all ranks evict, as the reference evicts one
        range in all layers."""
        draw = torch.rand(n)
        if mask_tail:
            draw[-mask_tail:] = -1e9
        e = int(torch.topk(draw, k=1, dim=-1)[1][0])
        if shard is not None:
            from . import dist as DS
            e = int(DS.broadcast_object(e, 0))
        return e

    if kv_mode == "auto":                                      # easykv/easykv.py:220-227
        assert type(budget) == int
        if budget > length:
            kv_mode, budget = "decoding", budget - length
        else:
            kv_mode = "encoding_decoding"

    # HF seam + streaming: the stock attention module hands over q/k already rotated by the TRUE positions, while the
    # streaming variant caches un-rotated keys and rotates by slot index on every read (llama_patch.py:310-327).  The
    # model's own rotary module supplies both the tables for the read-time rotation and the ones to take its rotation off.
    hf_stream = streaming and getattr(self.config, "_attn_implementation", None) == "easykv_amd"
    if streaming and not hf_stream:
        # native contract: the rotation at read time uses theta = config.rope_theta (Llama-3: 5e5, Mistral: 1e6); scaled
        # RoPE variants need the caller's own tables (generation_config['rope_tables'] = (cos, sin) fp32 [>= cap, D])
        rp = getattr(self.config, "rope_parameters", None) or {}
        scaling = getattr(self.config, "rope_scaling", None) or (rp if rp.get("rope_type", "default") != "default" else None)
        if scaling and cfg.get("rope_tables") is None:
            raise ValueError("streaming=True on a model with scaled RoPE needs generation_config['rope_tables'] = (cos, sin)")
    rope_base = float(getattr(self.config, "rope_theta", None) or (getattr(self.config, "rope_parameters", None) or {}).get("rope_theta", 10000.0))

    def new_cache(cap):
        hf_rope = cfg.get("rope_tables")
        if hf_stream:     # tables cover every slot index (< cap) and every true position (< length + max_new_tokens)
            from . import hf
            hf_rope = hf.rope_tables_from_model(self, max(cap + 8, length + max_new_tokens + 1) + 64, d, dev)
        cache = BudgetedKVCache(n_layers, hq, h, d, cap + 8, dev, streaming=streaming, record=record, rope=hf_rope,
                                layer_begin=l_begin, layer_count=l_count, rope_base=rope_base)
        cache.unrotate = hf_rope if hf_stream else None
        return cache

    def forward(cache, ids, positions, plan):
        plan.streaming = streaming
        pos = torch.as_tensor(positions, dtype=torch.long, device=dev)
        with cache.active(plan, pos):
            out = self(input_ids=ids, past_key_values=cache, position_ids=pos.view(1, -1), use_cache=True)
        # every owned layer must have gone through attend() exactly once: a model whose attention was not routed here
        # (an un-patched HF model) would otherwise silently run stock attention over the new rows only
        if cache.n_attend != cache.layer_count:
            raise RuntimeError(f"model forward made {cache.n_attend} attend() calls for {cache.layer_count} owned layers: route "
                               "every attention layer through past_key_values.attend (easykv_amd.hf.patch_model for HF models)")
        return out

    last_rank = shard.world - 1 if shard is not None else 0

    def sample(logits_last):
        """Next token [1, 1] on the device.  Sharded: only the last stage holds the logits; its draw is broadcast."""
        if shard is None:
            prob, raw = logits_adapter(logits_last.float(), temperature, top_p)
            return torch.multinomial(prob, num_samples=1)
        from . import dist as DS
        if shard.rank == last_rank:
            prob, raw = logits_adapter(logits_last.float(), temperature, top_p)
            tok = torch.multinomial(prob, num_samples=1)
        else:
            tok = torch.zeros(1, 1, dtype=torch.long, device=dev)
        return DS.broadcast(tok, last_rank)

    # extension key: the host looks at the sampled tokens every N tokens.  Default 1 = the reference's control flow (it tests
    # every token before feeding it, easykv/easykv.py:257-263); N > 1 is opt-in, see TokenLog
    eos_poll = max(1, int(cfg.get("eos_poll", 1)))

    class TokenLog:
        """Sampled tokens stay on the device.  The reference pulls every token to the host to test it for EOS
        (`.cpu()` / `.item()`, ~5 syncs per token, easykv/easykv.py:257-283); here the host polls the device-side log once per
        ``eos_poll`` tokens: ONE host sync per ``eos_poll`` tokens.  ``eos_poll=1`` (the default) is the reference's exact
        control flow: nothing runs past an EOS.  With N > 1 (opt-in) up to N-1 forwards run past an EOS before it is seen:
        the returned text and the printed budget line are still the reference's (cut at the first EOS; counts derived from
        the EOS index, not from the cache), but those forwards have evicted from the cache handed back by
        ``return_cache=True`` and have drawn from the sampler's / the 'random' policy's generators."""

        def __init__(self):
            self.buf = torch.empty(max(1, max_new_tokens), dtype=torch.long, device=dev)
            self.eos = torch.as_tensor([int(e) for e in eos_token_ids], dtype=torch.long, device=dev)
            self.n = self.checked = self.syncs = self.sampled = 0
            self.stopped_by_eos = False

        def push(self, tok):
            self.buf[self.n:self.n + 1].copy_(tok.view(1))
            self.n += 1
            self.sampled += 1      # every token drawn, including those past an EOS a later poll cuts off

        def poll(self):
            """True when the loop must stop: an EOS was found among the tokens not looked at yet (``n`` is cut back to it)."""
            if self.n - self.checked < eos_poll and self.n < max_new_tokens:
                return False
            hit = torch.isin(self.buf[self.checked:self.n], self.eos).cpu()     # the one host sync of this poll
            self.syncs += 1
            first = self.checked
            self.checked = self.n
            if bool(hit.any()):
                self.n = first + int(torch.nonzero(hit)[0, 0]) + 1
                self.stopped_by_eos = True
                return True
            return False

        def ids(self):
            return self.buf[:self.n].cpu().tolist()

        @property
        def fed(self):   # tokens the reference would have fed back into the model (:257-264: the EOS token itself is not)
            return self.n - 1 if self.stopped_by_eos else self.n

    class GraphedForward:
        """One forward of the WHOLE model captured in a hipGraph — a decode step (one token), or, since round 6, a
        strided chunk of the prefill (`stride` tokens).  At a fixed budget every evicting forward has the same shapes, the same
        StepPlan and the same cache length before and after (the cache oscillates idx <-> idx + stride, easykv/easykv.py:426-433), so
        the host work of a forward (HF's per-layer Python, ~0.4 ms per layer: 12 ms of a 13 ms chunk forward of a 32-layer model) is
        paid once at capture and a forward costs one graph launch.  Token ids and positions live in static device tensors; sampling, the
        EOS test and the collection of logits / evicted ids stay outside the graph."""

        def __init__(self, cache, plan, tok, positions):
            n = len(positions)
            self.tok = tok.view(1, n).clone()
            self.pos = torch.as_tensor(positions, dtype=torch.long).to(dev)
            self.graph = torch.cuda.CUDAGraph()
            plan.streaming = streaming
            self.cache = cache
            n_recorded = len(cache.evictions)
            tk = _ACTIVE.set(cache)
            try:
                with torch.cuda.graph(self.graph):      # capture launches nothing: the first replay runs this forward
                    cache.begin_forward(plan, self.pos)
                    self.logits = self_model(input_ids=self.tok, past_key_values=cache, position_ids=self.pos.view(1, -1),
                                             use_cache=True).logits
            finally:
                _ACTIVE.reset(tk)
            if cache.n_attend != cache.layer_count:
                raise RuntimeError(f"model forward made {cache.n_attend} attend() calls for {cache.layer_count} owned layers")
            # record=True: the captured forward left its (static) id tensors in the log; every replay appends a copy instead
            self.static_ids = cache.evictions.pop() if len(cache.evictions) > n_recorded else None
            self.layout = cache.bank.layout_signature()     # the captured kernels are those of THIS score-row layout

        def __call__(self, tok, positions):
            if self.cache.bank.layout_signature() != self.layout:
                raise RuntimeError("the bank's score-row layout changed between capture and replay of the forward's graph "
                                   "(an eager call on the bank in between): capture again")
            self.tok.copy_(tok.view(self.tok.shape))
            if len(positions) == 1:
                self.pos.fill_(positions[0])
            else:      # (consecutive positions, built on the device: no host-to-device copy on the replay path)
                torch.arange(positions[0], positions[0] + len(positions), dtype=torch.long, device=dev, out=self.pos)
            self.graph.replay()
            if self.static_ids is not None:
                self.cache.evictions.append([t.clone() for t in self.static_ids])
            return self.logits

    self_model = self

    # ---- single-token decode with eviction (decoding mode, and the tail of auto mode) ----------------------
    def decode_loop(cache, logits_last, cur_pos, score_off, budget_d, whole_cache):
        log = TokenLog()
        positions: List[int] = []
        graphed, prev_sig = None, None
        while log.n < max_new_tokens:                           # :257 / :670
            tok = sample(logits_last)
            log.push(tok)
            if log.poll():
                break
            t_now = cache.get_seq_length() + 1
            evict = evicting and (whole_cache or (t_now - score_off) > budget_d)     # :303 / every step :708
            plan = StepPlan(policy=policy, phase="decode", accumulate=scored, evict=evict, score_off=score_off, budget=budget_d)
            positions.append(cur_pos)
            if evict and policy in ("recency", "random"):
                if whole_cache:                                 # :741-747
                    if policy == "random":
                        raise UnboundLocalError("auto mode + kv_policy='random' is broken in the reference (easykv/easykv.py:744)")
                    plan.range_start = sink
                else:                                           # :343-362: oldest / uniformly random generated slot
                    # 'random': the reference's own draw — argmax of torch.rand over the generated slots, CPU generator
                    # (easykv/easykv.py:354-356), so a run seeded like a reference run evicts the same slots
                    e = 0 if policy == "recency" else policy_draw(len(positions))
                    positions.pop(e)
                    plan.range_start = score_off + e
            # steady state (same plan, same cache length as the step before, one slot evicted per step): replay the graph
            sig = (t_now, evict, plan.range_start)
            if use_graph and evict and policy != "random" and sig == prev_sig:
                if graphed is None:
                    graphed = GraphedForward(cache, plan, tok, [cur_pos])
                logits_last = graphed(tok, [cur_pos])[:, -1, :]
            else:
                logits_last = forward(cache, tok.view(1, 1), [cur_pos], plan).logits[:, -1, :]
            prev_sig = sig
            cur_pos += 1
        cache.host_syncs, cache.tokens_sampled = log.syncs, log.sampled
        return log.ids(), log.fed

    # ---- dense prefix + strided chunks with eviction (encoding, auto, ppl) ---------------------------------
    def prefill(cache, budget_p, idx, r_idx, tova_head_mean, keep_logits=False):
        recent = int(budget_p * recent_ratio)                    # :394
        cache.bank.state_init(idx + stride, 1 if keep_attention else 2, stride)       # :412-416
        cache.score_prefix = keep_attention
        # prefix [0, r_idx): dense causal; with keep_attention its probabilities seed S and Q (:396, :403-405)
        plan = StepPlan(policy="roco" if keep_attention else "full", phase="prefill", accumulate=keep_attention,
                        evict=False, stride=stride)
        # extension key `dense_growth` (default off = the reference's forward sequence): the chunks that only GROW the cache —
        # tokens [r_idx, idx): no eviction, and without keep_attention no accumulation either (easykv.py:443, :460) — attend
        # causally to everything before them, which is what the dense prefix does: they join it as ONE forward of idx tokens.
        # auto / ppl geometry takes the smallest r_idx (:551-552, :779-780), i.e. (idx - r_idx) / stride shape-changing forwards
        # that no graph can replay (256 of the 511 forwards of a 4096-token prompt at stride 8).  Same K / V rows, same state, same
        # evictions afterwards; with keep_attention the prefix' column sums are formed in one sweep instead of chunk by chunk
        # (equal up to fp32 summation order).
        r_dense = idx if cfg.get("dense_growth", False) else r_idx
        out = forward(cache, input_ids[:, :r_dense], list(range(r_dense)), plan)
        cache.score_prefix = False
        logits_last = out.logits[:, -1, :]
        all_logits, all_ids = [], []
        if keep_logits and r_dense > r_idx:      # (ppl mode collects the logits of every token from r_idx on, :816-901)
            all_logits.append(out.logits[0, r_idx:r_dense])
            all_ids.append(input_ids[0, r_idx:r_dense])
        r_idx = r_dense
        cur_pos = r_idx
        graphed, prev_sig = None, None
        for tok_i in range(r_idx, length, stride):                # :426
            t_now = cache.get_seq_length() + stride
            plan = StepPlan(policy=policy, phase="prefill", accumulate=scored and (t_now > idx or keep_attention),
                            evict=evicting and t_now > idx, budget=budget_p, recent=recent, sink=sink, stride=stride,
                            tova_head_mean=tova_head_mean)
            if plan.evict and policy == "recency":
                plan.range_start = sink                          # :491-493
            elif plan.evict and policy == "random":              # :494-499: argmax of torch.rand over the row, chunk excluded
                plan.range_start = policy_draw(idx + stride, stride)
            # steady state (generation_config['hipgraph']): from the second evicting chunk on every forward has the plan, the shapes and
            # the cache length of the one before it — captured once, replayed for the rest of the prompt
            sig = (t_now, plan.accumulate, plan.evict, plan.range_start)
            chunk_ids, chunk_pos = input_ids[:, tok_i:tok_i + stride], list(range(cur_pos, cur_pos + stride))
            if use_graph and plan.evict and policy != "random" and sig == prev_sig and chunk_ids.shape[1] == stride:
                if graphed is None:
                    graphed = GraphedForward(cache, plan, chunk_ids, chunk_pos)
                logits = graphed(chunk_ids, chunk_pos)
                if keep_logits:
                    logits = logits.clone()      # (the graph's output buffer is overwritten by the next replay)
            else:
                logits = forward(cache, chunk_ids, chunk_pos, plan).logits
            prev_sig = sig
            logits_last = logits[:, -1, :]
            if keep_logits:
                all_logits.append(logits[0])
                all_ids.append(input_ids[0, tok_i:tok_i + stride])
            cur_pos += stride
        cache.bank.release_workspace(keep_bytes=64 << 20)      # (deferred chunk steps keep all layers' logits / column sums: not the decode phase's business)
        return logits_last, all_logits, all_ids

    result = None
    if kv_mode == "decoding":                                     # easykv/easykv.py:228-366
        cap = length + (budget + 1 if evicting else max_new_tokens + 1)
        cache = new_cache(cap)
        out = forward(cache, input_ids, list(range(length)), StepPlan(policy="full", phase="prefill", accumulate=False))
        if evicting and scored:
            cache.bank.state_init(budget + 1, 0)                   # :242-245
        out_ids, fed = decode_loop(cache, out.logits[:, -1, :], length, length, budget, False)
        kept = min(fed, budget) if evicting else fed             # == cache length - prompt length when no forward ran past an EOS
        print(f"KV cache budget ratio: {kept / len(out_ids) * 100:.2f}%({kept}/{len(out_ids)})")
        result = self.tokenizer.decode(out_ids, skip_special_tokens=True).strip()

    elif kv_mode == "encoding":                                   # :367-529
        full = (type(budget) == float and budget >= 1.0) or (type(budget) == int and budget >= length)
        if full:
            cache = new_cache(length + max_new_tokens)
            logits_last = forward(cache, input_ids, list(range(length)),
                                  StepPlan(policy="full", phase="prefill", accumulate=False)).logits[:, -1, :]
        else:
            budget_p, idx, r_idx = geometry("encoding", length, budget, stride)
            # 'full' / unknown policy strings evict nothing (the reference's cache just grows): size for the whole prompt
            cache = new_cache((idx + stride if evicting else length) + max_new_tokens)
            logits_last, _, _ = prefill(cache, budget_p, idx, r_idx, True)
        kept = cache.get_seq_length()
        print(f"KV cache budget ratio: {kept / length * 100:.2f}%({kept}/{length})")
        log, cur_pos, t_first, n_fwd = TokenLog(), length, None, 0
        while log.n < max_new_tokens:                              # :508-526 plain decode, no eviction
            tok = sample(logits_last)
            log.push(tok)
            if log.poll():
                break
            logits_last = forward(cache, tok.view(1, 1), [cur_pos],
                                  StepPlan(policy="full", phase="decode", accumulate=False)).logits[:, -1, :]
            n_fwd += 1
            if report_decoding_latency and n_fwd == 1:             # the reference drops the first step from the mean (:527)
                torch.cuda.synchronize(dev)
                t_first = time.time()
            cur_pos += 1
        out_ids = log.ids()
        cache.host_syncs, cache.tokens_sampled = log.syncs, log.sampled
        result = self.tokenizer.decode(out_ids, skip_special_tokens=True).strip()
        if report_decoding_latency and n_fwd > 1:
            torch.cuda.synchronize(dev)
            print(f"Per-step decoding latency: {(time.time() - t_first) / (n_fwd - 1):.3f}")

    elif kv_mode == "encoding_decoding":                          # :530-753
        assert type(budget) == int and budget <= length
        white_lst = ["random", "recency", "tova", "roco"]
        assert policy in white_lst, f"mode must be within {white_lst}, get {policy} instead"
        assert stride > 1, "auto mode needs stride > 1 (the reference asserts at easykv/easykv.py:666-669)"
        budget_p, idx, r_idx = geometry("auto", length, budget, stride)
        cache = new_cache(idx + stride + 1)
        logits_last, _, _ = prefill(cache, budget_p, idx, r_idx, False)
        # the score rows keep their first idx+1 columns (:666-669); the decode rules then run over the whole cache
        out_ids, _ = decode_loop(cache, logits_last, length, 0, budget_p, True)
        size = cache.get_seq_length()
        print(f"KV Cache Budget ratio {size / (length + len(out_ids)) * 100:.2f}%[{size}/({length}+{len(out_ids)})]")
        result = self.tokenizer.decode(out_ids, skip_special_tokens=True).strip()

    elif kv_mode == "ppl":                                        # :754-901
        ce = torch.nn.CrossEntropyLoss(reduction="none")
        has_logits = shard is None or shard.rank == last_rank     # sharded: the logits exist on the last stage only
        if budget >= 1.0:     # NB: like the reference, ANY int budget takes this branch (:759); pass a ratio to evict
            cache = new_cache(length)
            out = forward(cache, input_ids, list(range(length)), StepPlan(policy="full", phase="prefill", accumulate=False))
            if has_logits:
                lp = ce(out.logits[0, :-1].float(), input_ids[0, 1:]).cpu().numpy().tolist()
                result = math.exp(statistics.mean(lp))
        else:
            budget_p, idx, r_idx = geometry("ppl", length, budget, stride)
            cache = new_cache(idx + stride if evicting else length)
            _, all_logits, all_ids = prefill(cache, budget_p, idx, r_idx, True, keep_logits=has_logits)
            kept = cache.get_seq_length()
            print(f"KV cache budget ratio: {kept / length * 100:.2f}%({kept}/{length})")
            if has_logits:
                ids_cat, log_cat = torch.cat(all_ids), torch.cat(all_logits, dim=0)
                assert ids_cat.shape[0] == log_cat.shape[0]
                lp = ce(log_cat[:-1].float(), ids_cat[1:]).cpu().numpy().tolist()
                result = math.exp(statistics.mean(lp))
        if shard is not None:
            from . import dist as DS
            result = DS.broadcast_object(result, last_rank)
    else:
        raise ValueError(f"unknown kv_mode {kv_mode!r}")
    if shard is not None:       # stage outputs still in flight (easykv_amd.dist.PipelineStage posts them without waiting)
        from . import dist as DS
        DS.drain_stages()
    return (result, cache) if return_cache else result


def enable_fixed_kv(model, tokenizer, mode, stride=1, verbose=False):
    """easykv/easykv.py:903-908."""
    model.tokenizer = tokenizer
    model.easykv_generate = functools.partial(generate, self=model, kv_mode=mode, stride=stride, report_decoding_latency=verbose)
    model.easykv_ppl = functools.partial(generate, self=model, kv_mode="ppl", stride=stride)
    print(f"Fixed KV Cache for {mode} enabled")
