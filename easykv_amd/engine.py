"""Device-resident budgeted KV bank driven through the C ABI (include/easykv_hip.h).

PyTorch is used for device memory and streams only; every operation on the bank is a HIP kernel
of ``libeasykv_hip.so``.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import Bank, Step, check

DECODE_RECENT_RATIO = 0.3   # easykv/easykv.py:308, :709 (hard-wired in the reference)
ROCO_TAIL = 10              # easykv/easykv.py:321, :472, :721


@dataclass
class StepPlan:
    """What the driver decides for one model forward; identical for every layer (mirrors the
    branches of easykv/easykv.py:287-362 for ``phase='decode'`` and :443-499 for ``'prefill'``)."""
    policy: str = "roco"
    phase: str = "decode"
    accumulate: bool = True
    evict: bool = False
    score_off: int = 0          # P: first position covered by the score rows
    budget: int = 0             # decode: budget ; prefill: budget' (= budget + stride)
    recent: int = 0             # prefill: int(budget' * recent_ratio)
    sink: int = 0               # prefill: temp_length
    stride: int = 1
    tova_head_mean: bool = False
    range_start: int = -1       # recency / random
    streaming: bool = False
    n_split: int = 0
    two_pass: int = 0           # scored chunk steps: 0 auto, 1 force the two-pass kernels, -1 force one pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _row_strides(t: torch.Tensor):
    """(token_stride, head_stride) in elements when the rows of ``t`` ``[layers, heads, n, D]`` can be read in place by the kernels
    (ekv_step.*_stride, ABI 8): fp16, D contiguous halfs per row, layers dense blocks apart, strides multiples of 8 — e.g. the
    ``[1, heads, n, D]`` transposed views HF attention modules hand over.  (0, 0) = the dense layout; None = needs a copy."""
    layers, heads, n, d = t.shape
    s0, s1, s2, s3 = t.stride()
    if t.dtype != torch.float16 or s3 != 1 or (t.data_ptr() & 15):
        return None
    if layers > 1 and s0 != heads * n * d:
        return None
    ts = s2 if n > 1 else d
    hs = s1 if heads > 1 else max(n * d, ts)
    if ts == d and hs == n * d:
        return 0, 0
    if n == 1:
        return (0, 0) if hs == d else None
    if ts < d or hs < d or (ts & 7) or (hs & 7):
        return None
    return ts, hs


def _stride_rows(st, q, k_new, v_new, out):
    """Fill the row strides of ``st`` from the tensors; tensors whose layout the kernels cannot read in place are copied dense (k_new
    and v_new share one pair of strides).  Returns (q, k_new, v_new)."""
    f16 = torch.float16
    if (q.dtype is f16 and k_new.dtype is f16 and v_new.dtype is f16 and q.is_contiguous() and k_new.is_contiguous() and v_new.is_contiguous()
            and (out is None or out.is_contiguous())):
        # dense tensors (the bank-level callers: tests, bench): a few C++ calls — this sits on the per-layer critical path of a decoder
        # stack, where the host cost of a call is what bounds small layers
        if st.q_token_stride or st.q_head_stride or st.kv_token_stride or st.kv_head_stride or st.out_token_stride or st.out_head_stride:
            st.q_token_stride = st.q_head_stride = st.kv_token_stride = st.kv_head_stride = st.out_token_stride = st.out_head_stride = 0
        return q, k_new, v_new
    sq = _row_strides(q)
    if sq is None:
        q, sq = q.to(torch.float16).contiguous(), (0, 0)
    sk, sv = _row_strides(k_new), _row_strides(v_new)
    if sk is None or sv != sk:
        k_new, v_new, sk = k_new.to(torch.float16).contiguous(), v_new.to(torch.float16).contiguous(), (0, 0)
    so = (0, 0) if out is None else _row_strides(out)
    if so is None:
        raise ValueError("`out` must be an fp16 tensor whose rows are head_dim contiguous halfs at strides that are multiples of 8")
    (st.q_token_stride, st.q_head_stride), (st.kv_token_stride, st.kv_head_stride), (st.out_token_stride, st.out_head_stride) = sq, sk, so
    return q, k_new, v_new


class KVBank:
    """K/V rows + slot map + score rows of ``n_layers`` layers on one GPU."""

    default_two_pass = 0    # StepPlan.two_pass used when a plan leaves it at 0 (tests force either chunk scheme with it)

    def __init__(self, n_layers, n_q_heads, n_kv_heads, head_dim, cap, device="cuda", scored=True):
        self.lib = _lib.load()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.EkvError("KVBank needs a GPU device; the product path has no CPU fallback")
        self.device = dev
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        cap = (cap + 63) // 64 * 64     # rows of the slot map / score rows stay 16-byte aligned (fused kernel, LDS-DMA)
        self.n_layers, self.n_q_heads, self.n_kv_heads, self.head_dim, self.cap = n_layers, n_q_heads, n_kv_heads, head_dim, cap
        self.k = torch.empty(n_layers, n_kv_heads, cap, head_dim, dtype=torch.float16, device=dev)
        self.v = torch.empty_like(self.k)
        # (the state tensors are reached through properties: reading one converts slot-indexed layers back to the ordered layout)
        self._slot_of_pos = torch.empty(n_layers, n_kv_heads, cap, dtype=torch.int32, device=dev)
        self._score_sum = torch.zeros(n_layers, n_kv_heads, cap, dtype=torch.float32, device=dev) if scored else None
        self._score_sq = torch.zeros_like(self._score_sum) if scored else None
        self._score_cnt = torch.zeros_like(self._score_sum) if scored else None
        # slot-indexed score rows (include/easykv_hip.h, EKV_PHASE_SLOT_ROWS): order keys, (count offset, next birth) per head, and
        # which layers currently hold that layout
        self.birth = torch.zeros(2, n_layers, n_kv_heads, cap, dtype=torch.int32, device=dev) if scored else None
        self.slot_state = torch.zeros(n_layers, n_kv_heads, 4, dtype=torch.float32, device=dev) if scored else None
        self._slot_rows = [False] * n_layers
        self._slot_min_tail = [1 << 30] * n_layers      # smallest protected tail of an evicting step since the conversion
        self._slot_ok = {}        # step shape -> does the slot-indexed decode kernel take it (ekv_step_check)
        self._slot_stretch = self._slot_short = 0      # steps since the layout was entered; short stretches seen (see _ensure_ordered)
        self._slot_cool = 0       # one-launch decode steps run on the ordered layout since the thrash guard closed (see _enter_slot_rows)
        self.n_slots = [0] * n_layers
        # High-water mark of live rows per layer.  The library keeps the free list as [freed rows (most recent first), never-
        # used rows ascending] and appends take from its front, so every live row has a physical index < extent: the
        # one-launch decode step streams the rows [0, extent) in address order (ekv_step.phys_extent).
        self.extent = [0] * n_layers
        self._ws = None
        # scorer off the critical path (attend(..., overlap_scorer=True)): side streams, a ring of workspaces and, per
        # layer, the event after which its slot map / score rows are up to date again
        self._side = None
        self._ws_ring = []
        self._ws_free = []
        self._ring_pos = 0
        self._score_done = [None] * n_layers
        self._defer = None      # deferred-scorer state of the token step in flight (attend(..., defer=True) ... flush())
        self.arrive = torch.zeros(n_layers, n_kv_heads, dtype=torch.int32, device=dev)     # arrival counters of the in-kernel fold
        self._bank = Bank(self.k.data_ptr(), self.v.data_ptr(), self._slot_of_pos.data_ptr(),
                          self._score_sum.data_ptr() if scored else None, self._score_sq.data_ptr() if scored else None,
                          self._score_cnt.data_ptr() if scored else None, n_layers, n_q_heads, n_kv_heads, head_dim, cap,
                          self.arrive.data_ptr(), self.birth.data_ptr() if scored else None,
                          self.slot_state.data_ptr() if scored else None)
        self.rope_cos = self.rope_sin = None
        self.reset()

    # -- layout of the score rows -------------------------------------------------------------------
    use_slot_rows = True      # one-launch decode steps run on the slot-indexed layout where the library supports it
    SLOT_COOL_DOWN = 256      # ordered one-launch decode steps after which a bank whose thrash guard closed tries the layout again

    def layout_signature(self):
        """Which layers hold the slot-indexed layout.  A hipGraph captured over steps of this bank replays the kernels of the layout
        it was captured on: whoever replays one must find the same signature as at capture time (an eager call in between — a read
        of ``bank.score_sum``, a chunk step — converts layers back; only the conversion itself is refused DURING a capture)."""
        return tuple(self._slot_rows)

    def _ensure_ordered(self, layer_begin=0, layer_count=None):
        """Bring the score rows / slot map of the layers back to the ordered layout (every kernel but the one-launch decode step
        reads that one; so do the state tensors handed out below)."""
        end = self.n_layers if layer_count is None else layer_begin + layer_count
        l = layer_begin
        while l < end:
            if not self._slot_rows[l]:
                l += 1
                continue
            m = l
            while m < end and self._slot_rows[m] and self.n_slots[m] == self.n_slots[l]:
                m += 1
            if torch.cuda.is_current_stream_capturing():
                raise _lib.EkvError("a layout change of the score rows cannot be captured in a graph: run the step once outside the capture")
            check(self.lib.ekv_rows_to_order(C.byref(self._bank), l, m - l, self.n_slots[l], self._stream()), "ekv_rows_to_order")
            # a caller that alternates between one-launch decode steps and anything that reads the ordered layout pays two
            # conversions per step (~0.2 ms per 256 heads each): after a few stretches of under 16 steps the bank stays ordered
            self._slot_short = self._slot_short + 1 if self._slot_stretch < 16 else max(0, self._slot_short - 1)
            self._slot_stretch = 0
            for i in range(l, m):
                self._slot_rows[i] = False
            l = m

    def _enter_slot_rows(self, st) -> bool:
        """Decide whether this one-launch decode step runs on the slot-indexed layout, converting its layers if it does."""
        lb, lc = st.layer_begin, st.layer_count
        rows = self._slot_rows[lb:lb + lc]
        if not (self.use_slot_rows and self._score_sum is not None) or len(set(self.n_slots[lb:lb + lc])) != 1:
            return False
        if self._slot_short >= 4:
            # thrash guard (see _ensure_ordered): the caller kept interleaving state reads with decode steps.  Not for ever — a
            # generate that reads the state during its first tokens and then settles into pure decode gets the layout back after a
            # cool-down of ordered steps; one more short stretch closes the guard again at once
            self._slot_cool += 1
            if self._slot_cool < self.SLOT_COOL_DOWN or torch.cuda.is_current_stream_capturing():
                return False
            self._slot_short, self._slot_cool = 3, 0
        st.phases = _lib.PHASE_SLOT_ROWS
        key = bytes(st)
        ok = self._slot_ok.get(key)
        if ok is None:
            if len(self._slot_ok) > 4096:      # (a growing cache asks about a new shape every step)
                self._slot_ok.clear()
            ok = self._slot_ok[key] = self.lib.ekv_step_check(C.byref(self._bank), C.byref(st)) == 0
        st.phases = 0
        if not ok:
            return False
        if not all(rows):
            if torch.cuda.is_current_stream_capturing():
                return False          # (no layout change inside a capture: this step runs on the ordered layout)
            self._ensure_ordered(lb, lc)     # (a partly converted range)
            check(self.lib.ekv_rows_to_slots(C.byref(self._bank), lb, lc, self.n_slots[lb], self._stream()), "ekv_rows_to_slots")
            for i in range(lb, lb + lc):
                self._slot_rows[i] = True
                self._slot_min_tail[i] = 1 << 30      # (births = order indices: every tail is consecutive)
        self._slot_stretch += 1
        return True

    @property
    def slot_of_pos(self):
        self._ensure_ordered()
        return self._slot_of_pos

    @property
    def score_sum(self):
        self._ensure_ordered()
        return self._score_sum

    @property
    def score_sq(self):
        self._ensure_ordered()
        return self._score_sq

    @property
    def score_cnt(self):
        self._ensure_ordered()
        return self._score_cnt

    # -- plumbing -----------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._dev_index).cuda_stream)

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
        return self._ws

    def release_workspace(self, keep_bytes=0):
        """Drop the cached scratch buffer when it is larger than ``keep_bytes`` (a deferred chunk step of a long prefill keeps every
        layer's logits / column sums alive until the flush: up to GBs that the decode phase after it never needs)."""
        if self._ws is not None and self._ws.numel() > keep_bytes and (self._defer is None or not self._defer["pending"]):
            self._ws = None
            self._defer = None      # (its cached pointers referred to the buffer)

    def reset(self):
        check(self.lib.ekv_bank_reset(C.byref(self._bank), self._stream()), "ekv_bank_reset")
        self.n_slots = [0] * self.n_layers
        self.extent = [0] * self.n_layers
        self._slot_rows = [False] * self.n_layers
        self._slot_min_tail = [1 << 30] * self.n_layers
        self._slot_stretch = self._slot_short = self._slot_cool = 0
        self._defer = None      # (a half-open deferred token step dies with the bank's contents; arrive[] is zeroed by the reset)

    def abort_step(self):
        """Drop a deferred token step that was opened (``attend(..., defer=True)`` on some layers) but never flushed — a model
        forward that raised half-way through the stack.  The layers that did attend have written the new token's row into the
        slot their free list pointed at, but ``n_slots`` has not advanced, so the next step of every layer reuses that very
        slot: the bank is as it was before the aborted token.  Returns the number of layers that had attended."""
        d, n = self._defer, 0
        if d is not None:
            n, d["pending"], d["t"] = d["pending"], 0, -1
            if n:   # a launch that died mid-way could leave arrival counters of the in-kernel fold non-zero
                self.arrive.zero_()
        return n

    def state_init(self, width, mode, stride=1, layer_begin=0, layer_count=None):
        """mode 0 decoding (easykv/easykv.py:242-245); 1 prefill+keep_attention; 2 prefill (:412-416)."""
        lc = self.n_layers - layer_begin if layer_count is None else layer_count
        self._ensure_ordered(layer_begin, lc)      # (the slot map is not rewritten by the initialisation)
        check(self.lib.ekv_state_init(C.byref(self._bank), layer_begin, lc, width, mode, stride, self._stream()), "ekv_state_init")

    def set_rope(self, cos, sin):
        """fp32 tables ``[>= cap, head_dim]`` for the streaming (rope-on-read) variant."""
        self.rope_cos = cos.to(self.device, torch.float32).contiguous()
        self.rope_sin = sin.to(self.device, torch.float32).contiguous()
        half = self.head_dim // 2
        # the kernels read only the first half of a table row for both halves of the head (the reference's tables are
        # cat(freqs, freqs), llama_patch.py:74-98 / HF rotary modules)
        if not (torch.equal(self.rope_cos[:, :half], self.rope_cos[:, half:]) and torch.equal(self.rope_sin[:, :half], self.rope_sin[:, half:])):
            raise _lib.EkvError("rope tables must have the cat(freqs, freqs) layout (equal halves along head_dim)")
        # ... and must be a rotation that is linear in the position, row j = (cos(j * theta_f), sin(j * theta_f)): the decode stream
        # reads one row per run of consecutive positions and advances (cos, sin) by the angle-addition recurrence with row 1 as the
        # step (include/easykv_hip.h).  Every RoPE variant is (linear / NTK / llama3 / yarn only change theta_f); the tables' own fp32
        # rounding of the angle (6e-8 * j * theta_f) is the only deviation allowed for.
        if self.rope_cos.shape[0] > 2:
            c, s_ = self.rope_cos[:, :half].double(), self.rope_sin[:, :half].double()
            dev_c = (c[:-1] * c[1] - s_[:-1] * s_[1] - c[1:]).abs().max()
            dev_s = (s_[:-1] * c[1] + c[:-1] * s_[1] - s_[1:]).abs().max()
            if float(torch.maximum(dev_c, dev_s)) > 1e-4 + 4e-7 * self.rope_cos.shape[0]:
                raise _lib.EkvError("rope tables must be rotations linear in the position: row j = (cos(j * theta), sin(j * theta)) per frequency")

    # -- data movement at the boundary -------------------------------------------------------------
    def load_rows(self, k, v, pos_begin=None, layer_begin=0):
        """Append ordered rows ``[layers, H, n, D]`` at positions [pos_begin, pos_begin+n)."""
        lc, n = k.shape[0], k.shape[2]
        self._ensure_ordered(layer_begin, lc)
        pos = self.n_slots[layer_begin] if pos_begin is None else pos_begin
        k = k.to(self.device, torch.float16).contiguous()
        v = v.to(self.device, torch.float16).contiguous()
        check(self.lib.ekv_scatter_rows(C.byref(self._bank), layer_begin, lc, pos, n, _ptr(k), _ptr(v), self._stream()), "ekv_scatter_rows")
        for l in range(layer_begin, layer_begin + lc):
            self.n_slots[l] = pos + n
            self.extent[l] = max(self.extent[l], pos + n)

    def ordered_kv(self, layer_begin=0, layer_count=None):
        """The ordered ``[layers, H, T, D]`` view the HF legacy tuple needs (birth order)."""
        lc = self.n_layers - layer_begin if layer_count is None else layer_count
        self._ensure_ordered(layer_begin, lc)
        t = self.n_slots[layer_begin]
        k = torch.empty(lc, self.n_kv_heads, t, self.head_dim, dtype=torch.float16, device=self.device)
        v = torch.empty_like(k)
        check(self.lib.ekv_gather_ordered(C.byref(self._bank), layer_begin, lc, t, _ptr(k), _ptr(v), self._stream()), "ekv_gather_ordered")
        return k, v

    def compact_inplace(self, evict_ids, layer_begin=0):
        """Reference-shaped physical compaction for a bank kept in identity layout."""
        lc, _, kk = evict_ids.shape
        self._ensure_ordered(layer_begin, lc)
        t = self.n_slots[layer_begin]
        ids = evict_ids.to(self.device, torch.int32).contiguous()
        check(self.lib.ekv_compact_inplace(C.byref(self._bank), layer_begin, lc, t, kk, _ptr(ids), self._stream()), "ekv_compact_inplace")
        for l in range(layer_begin, layer_begin + lc):
            self.n_slots[l] = t - kk

    # -- the fused step -----------------------------------------------------------------------------
    def make_step(self, plan: StepPlan, q_len, layer_begin, layer_count) -> Step:
        t = self.n_slots[layer_begin] + q_len
        st = Step()
        st.layer_begin, st.layer_count, st.q_len, st.n_slots = layer_begin, layer_count, q_len, t
        st.score_off = plan.score_off
        st.policy = _lib.POLICY_CODES.get(plan.policy, _lib.POLICY_NONE)   # unknown string = no-op, as in the reference
        st.accumulate = int(plan.accumulate)
        st.causal = 1
        st.rope_on_read = int(plan.streaming)
        st.n_split = plan.n_split
        st.two_pass = plan.two_pass or KVBank.default_two_pass
        st.sm_div = math.sqrt(self.head_dim)
        st.tova_head_mean = int(plan.tova_head_mean)
        st.roco_tail = ROCO_TAIL
        st.range_start = -1
        st.phys_extent = max(max(self.extent[layer_begin:layer_begin + layer_count]), t)
        if plan.evict and st.policy != _lib.POLICY_NONE:
            if plan.phase == "decode":
                rw = int(plan.budget * DECODE_RECENT_RATIO)
                st.n_evict = 1
                st.win_lo, st.win_tail = (0, rw) if plan.policy == "h2o_head" else (0, 0)
                st.roco_k1 = plan.budget - rw
                st.count_add, st.count_tail_step = 1.0, 0.0
            else:
                st.n_evict = plan.stride
                st.win_lo, st.win_tail = plan.sink, plan.recent
                st.roco_k1 = max(plan.budget - plan.recent - plan.sink, plan.stride)
                st.count_add, st.count_tail_step = float(plan.stride), -1.0
            if st.policy == _lib.POLICY_RANGE:
                st.range_start = plan.range_start
        return st

    def step_plan(self, plan: StepPlan, q_len, layer_begin=0, layer_count=None, phases=0):
        """(n_split, fused) the library will use for this step."""
        st = self.make_step(plan, q_len, layer_begin, self.n_layers - layer_begin if layer_count is None else layer_count)
        st.phases = phases
        ns, fu = C.c_int32(0), C.c_int32(0)
        check(self.lib.ekv_step_plan(C.byref(self._bank), C.byref(st), C.byref(ns), C.byref(fu)), "ekv_step_plan")
        return ns.value, bool(fu.value)

    def step_info(self, plan: StepPlan, q_len, layer_begin=0, layer_count=None, phases=0) -> dict:
        """The library's dispatch decisions for this step (ekv_step_info): n_split, fused, two_pass, wide, n_qblocks, ..."""
        st = self.make_step(plan, q_len, layer_begin, self.n_layers - layer_begin if layer_count is None else layer_count)
        st.phases = phases
        info = (C.c_int32 * 9)()
        check(self.lib.ekv_step_info(C.byref(self._bank), C.byref(st), info, 9), "ekv_step_info")
        keys = ("n_split", "fused", "two_pass", "wide", "n_qblocks", "qb_rows", "n_col_parts", "fold_in_kernel", "n_launches")
        return dict(zip(keys, (int(x) for x in info)))

    def join(self):
        """Make the current stream wait for every scorer still running on a side stream."""
        cur = torch.cuda.current_stream(self.device)
        for l, ev in enumerate(self._score_done):
            if ev is not None:
                cur.wait_event(ev)
                self._score_done[l] = None
        self._ws_free = [None] * len(self._ws_free)     # every scorer joined: all workspace slots are free again

    def _attend_overlapped(self, st, q, k_new, v_new, out, evict_ids, lc, layer_begin):
        """Attention + fold on the current stream (the attention output is ready as early as possible); the scorer
        (accumulate / select / compaction) on a side stream.  The next attention of these layers waits for it."""
        if self._side is None:
            self._side = [torch.cuda.Stream(self.device) for _ in range(4)]
        main = torch.cuda.current_stream(self.device)
        need = self.lib.ekv_workspace_bytes(C.byref(self._bank), C.byref(st))
        if not self._ws_ring or self._ws_ring[0].numel() < need:
            self._ws_ring = [torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.device) for _ in range(8)]
            self._ws_free = [None] * 8
        slot = self._ring_pos
        self._ring_pos = (slot + 1) % len(self._ws_ring)
        ws = self._ws_ring[slot]
        for ev in [self._ws_free[slot]] + self._score_done[layer_begin:layer_begin + lc]:
            if ev is not None:
                main.wait_event(ev)      # workspace slot free again; slot map / score rows of these layers up to date
        args = (_ptr(q), _ptr(k_new), _ptr(v_new), _ptr(out), _ptr(evict_ids) if st.n_evict > 0 else None,
                _ptr(self.rope_cos), _ptr(self.rope_sin), _ptr(ws), ws.numel())
        st.phases = 1 | 4
        check(self.lib.ekv_step_attend(C.byref(self._bank), C.byref(st), *args, C.c_void_p(main.cuda_stream)), "ekv_step_attend")
        ready = torch.cuda.Event()
        ready.record(main)
        side = self._side[slot % len(self._side)]
        side.wait_event(ready)
        st.phases = 8
        check(self.lib.ekv_step_attend(C.byref(self._bank), C.byref(st), *args, C.c_void_p(side.cuda_stream)), "ekv_step_attend")
        done = torch.cuda.Event()
        done.record(side)
        self._ws_free[slot] = done
        for l in range(layer_begin, layer_begin + lc):
            self._score_done[l] = done

    # -- one layer per call (a real decoder stack): attention now, scoring / eviction once per token ------------------------
    def _attend_deferred(self, plan: StepPlan, q, k_new, v_new, layer, out):
        """Step of ONE layer with the scorer deferred (ekv_step.defer_layers): attention + fold of this layer now — its output is
        what the next layer waits for — and one scorer launch over all layers of the bank at :meth:`flush`, instead of a
        latency-bound scorer kernel of 32 workgroups on the critical path of every layer (decode: 15 us of a 25 us layer; a
        96-row chunk step of the Llama2-7B shape: 66 us of a 124 us layer).  Decode steps and, since round 4, chunk steps."""
        d = self._defer
        n = q.shape[2]
        t = self.n_slots[layer] + n
        if self._slot_rows[layer]:
            self._ensure_ordered()      # (all layers in one launch: the conversion ranks births by counting, ~0.2 ms per launch of <= 256 heads)
        if d is None or d["plan"] is not plan or d["t"] != t or d["n"] != n:
            if d is not None and d["pending"]:
                raise _lib.EkvError("attend(defer=True): the previous step was not flushed")
            st = self.make_step(plan, n, 0, 1)
            if plan.n_split <= 0:      # the split count of a one-layer launch, fixed for both halves of the step
                st.phases = 1
                ns, fu = C.c_int32(0), C.c_int32(0)
                check(self.lib.ekv_step_plan(C.byref(self._bank), C.byref(st), C.byref(ns), C.byref(fu)), "ekv_step_plan")
                st.n_split = ns.value
            st.defer_layers = self.n_layers
            # the scorer shape of the flush() call (phases = 8 over all layers) is validated NOW, before the first layer's
            # attention appends a row: a shape only the last call of the token would refuse must not leave the bank half-stepped
            st.layer_begin, st.layer_count, st.defer_index, st.phases = 0, self.n_layers, 0, 8
            check(self.lib.ekv_step_check(C.byref(self._bank), C.byref(st)), "ekv_step_check (deferred scorer)")
            need = self.lib.ekv_workspace_bytes(C.byref(self._bank), C.byref(st))
            ids = torch.empty(self.n_layers, self.n_kv_heads, st.n_evict, dtype=torch.int32, device=self.device) if st.n_evict > 0 else None
            ws = self._workspace(need)
            # everything that is the same for all layers of the token step is resolved once: the per-layer call below is on the
            # critical path of the decoder stack (host cost per layer ~ GPU cost per layer in this regime)
            d = self._defer = dict(plan=plan, t=t, n=n, st=st, ws=ws, ids=ids, pending=0, rope=(_ptr(self.rope_cos), _ptr(self.rope_sin)),
                                   st_ref=C.byref(st), bank_ref=C.byref(self._bank), ws_ptr=ws.data_ptr(), ws_len=ws.numel(),
                                   stream=self._stream(), call=self.lib.ekv_step_attend)
        st = d["st"]
        st.layer_begin = st.defer_index = layer
        st.layer_count, st.phases = 1, 5
        ext = self.extent[layer]
        st.phys_extent = ext if ext > t else t
        if out is None:
            out = torch.empty(1, self.n_q_heads, n, self.head_dim, dtype=torch.float16, device=self.device)
        q, k_new, v_new = _stride_rows(st, q, k_new, v_new, out)
        rc = d["call"](d["bank_ref"], d["st_ref"], q.data_ptr(), k_new.data_ptr(), v_new.data_ptr(), out.data_ptr(), None,
                       d["rope"][0], d["rope"][1], d["ws_ptr"], d["ws_len"], d["stream"])
        if rc != 0:
            check(rc, "ekv_step_attend")
        self.extent[layer] = st.phys_extent
        d["pending"] += 1
        return out

    def deferred_workspace_bytes(self, plan: StepPlan, q_len: int) -> int:
        """Scratch a step of ``q_len`` queries needs when the scorers of all layers are deferred to :meth:`flush` (the logits / column
        sums of every layer stay alive until then)."""
        st = self.make_step(plan, q_len, 0, 1)
        if plan.n_split <= 0:
            st.phases = 1
            ns, fu = C.c_int32(0), C.c_int32(0)
            check(self.lib.ekv_step_plan(C.byref(self._bank), C.byref(st), C.byref(ns), C.byref(fu)), "ekv_step_plan")
            st.n_split = ns.value
        st.defer_layers, st.layer_begin, st.layer_count, st.defer_index, st.phases = self.n_layers, 0, self.n_layers, 0, 8
        return int(self.lib.ekv_workspace_bytes(C.byref(self._bank), C.byref(st)))

    def flush(self):
        """Scorer of every layer of the token step opened by ``attend(..., defer=True)``: accumulate, select, compact — one
        launch.  Returns the evicted positions ``[layers, H, 1]`` (or None)."""
        d = self._defer
        if d is None or d["pending"] == 0:
            return None
        if d["pending"] != self.n_layers:
            raise _lib.EkvError(f"flush(): {d['pending']} of {self.n_layers} layers attended in this token step")
        st, ws = d["st"], d["ws"]
        st.layer_begin, st.layer_count, st.defer_index, st.phases = 0, self.n_layers, 0, 8
        st.q_token_stride = st.q_head_stride = st.kv_token_stride = st.kv_head_stride = st.out_token_stride = st.out_head_stride = 0      # (no caller tensors in this call)
        st.phys_extent = max(max(self.extent), d["t"])
        check(self.lib.ekv_step_attend(C.byref(self._bank), C.byref(st), ws.data_ptr(), ws.data_ptr(), ws.data_ptr(), ws.data_ptr(),
                                       _ptr(d["ids"]), d["rope"][0], d["rope"][1], ws.data_ptr(), ws.numel(), d["stream"]), "ekv_step_attend")
        for l in range(self.n_layers):
            self.n_slots[l] = d["t"] - st.n_evict
        d["pending"] = 0
        d["t"] = -1
        return d["ids"] if st.n_evict > 0 else None

    def attend(self, plan: StepPlan, q, k_new, v_new, layer_begin=0, out=None, evict_ids=None, phases=0, overlap_scorer=False, defer=False):
        """q ``[layers, Hq, n, D]``, k_new/v_new ``[layers, H, n, D]`` (fp16, device; dense or strided views whose rows are read in
        place — ekv_step.*_stride, see :func:`_row_strides`; anything else is copied dense first).
        Returns (out ``[layers, Hq, n, D]`` fp16, evict_ids ``[layers, H, k]`` int32 or None).
        ``defer=True`` (one layer per call): attention + fold only; the scorers of all layers run at :meth:`flush`.
        ``evict_ids=False``: the caller has no use for the evicted cache indices (none are returned; on the slot-indexed layout the
        step then skips ranking the victim's birth)."""
        if defer:
            if q.shape[0] != 1 or phases != 0:
                raise ValueError("defer=True is for one-layer calls")
            return self._attend_deferred(plan, q, k_new, v_new, layer_begin, out), None
        lc, _, n, _ = q.shape
        st = self.make_step(plan, n, layer_begin, lc)
        # one-launch decode steps run on the slot-indexed layout of the score rows (nothing moves on an eviction); everything else
        # on the ordered one
        slot = n == 1 and phases == 0 and not overlap_scorer and self._enter_slot_rows(st)
        if not slot and any(self._slot_rows[layer_begin:layer_begin + lc]):
            self._ensure_ordered()      # (all layers in one launch: a layer-per-call caller would otherwise convert 32 times)
        st.phases = phases | (_lib.PHASE_SLOT_ROWS if slot else 0)
        if slot and st.n_evict > 0:
            roco = st.policy == _lib.POLICY_ROCO
            tail = st.roco_tail if roco else st.win_tail
            # roco protects its newest entries by std = 1e9 sentinels only (easykv/easykv.py:318-321): when the feasible set has to
            # reach into them (k1 > T - tail: budgets below ~30) the arg-min over the mean may evict one of the newest `tail`
            # entries, and their births are no longer consecutive.  Such a step — and every step after it, until the next
            # conversion — leaves the proof to the kernel's counting check (exact bisection when it fails).
            breaks_tail = roco and st.roco_k1 > st.n_slots - st.roco_tail
            if not breaks_tail and all(tail <= self._slot_min_tail[l] for l in range(layer_begin, layer_begin + lc)):
                st.phases |= _lib.PHASE_SLOT_TAIL_OK
            for l in range(layer_begin, layer_begin + lc):
                self._slot_min_tail[l] = 0 if breaks_tail else min(self._slot_min_tail[l], tail)
        if out is None:
            out = torch.empty(lc, self.n_q_heads, n, self.head_dim, dtype=torch.float16, device=self.device)
        q, k_new, v_new = _stride_rows(st, q, k_new, v_new, out)
        want_ids = evict_ids is not False
        if evict_ids is False:      # the caller has no use for the evicted order indices (the slot-indexed layout then skips ranking the victim)
            evict_ids = None if slot else torch.empty(lc, self.n_kv_heads, max(st.n_evict, 1), dtype=torch.int32, device=self.device)
        elif st.n_evict > 0 and evict_ids is None:
            evict_ids = torch.empty(lc, self.n_kv_heads, st.n_evict, dtype=torch.int32, device=self.device)
        if overlap_scorer and phases == 0 and not self.step_plan(plan, n, layer_begin, lc)[1]:
            self._attend_overlapped(st, q, k_new, v_new, out, evict_ids, lc, layer_begin)
            for l in range(layer_begin, layer_begin + lc):
                self.n_slots[l] = st.n_slots - st.n_evict
                self.extent[l] = st.phys_extent
            return out, (evict_ids if (st.n_evict > 0 and want_ids) else None)
        if any(ev is not None for ev in self._score_done[layer_begin:layer_begin + lc]):
            self.join()
        need = self.lib.ekv_workspace_bytes(C.byref(self._bank), C.byref(st))
        ws = self._workspace(need)
        check(self.lib.ekv_step_attend(C.byref(self._bank), C.byref(st), _ptr(q), _ptr(k_new), _ptr(v_new), _ptr(out),
                                       _ptr(evict_ids) if (st.n_evict > 0 and evict_ids is not None) else None, _ptr(self.rope_cos), _ptr(self.rope_sin),
                                       _ptr(ws), ws.numel(), self._stream()), "ekv_step_attend")
        if phases != 1:     # phases == 1 launches the attention kernel only; the slot map is untouched
            for l in range(layer_begin, layer_begin + lc):
                self.n_slots[l] = st.n_slots - st.n_evict
        for l in range(layer_begin, layer_begin + lc):      # the new rows are written by the attention kernel
            self.extent[l] = st.phys_extent
        return out, (evict_ids if (st.n_evict > 0 and want_ids) else None)
