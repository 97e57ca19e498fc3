#!/usr/bin/env python
"""Bench-D: decode at fixed budget over the budgeted-KV attention path.

One "step" = one decode token of the path: for each of the L=32 layers of the Llama2-7B shape
(B=1, Hq=H=32, D=128) a fused HIP step — append the new K/V row, attention of the query over the
T = budget+1 = 2049 retained slots, score accumulation (roco: sum p, sum p^2, count), victim
selection and slot-map/score-row compaction — so the cache stays at `budget` slots.  Inputs are
synthetic N(0,1) fp16 (resident in HBM before the timed region), weights do not exist on this
path.  The metric is BASELINE.json's: decode tokens/s (path only) + HBM GB/s of the kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by torch.distributed.run, one rank per GPU.  Eviction state is per (layer, head), so layers
shard as contiguous blocks with no data-path collective.  `--scaling strong` (default): the ONE
Llama2-7B-shaped model is split over the ranks, 32/N layers each — a full pipeline, every stage busy with a different
sequence's token each step; the stage's real output ([1, Hq*D] fp16, the attention output of its last layer) goes to
the next rank by an RCCL point-to-point pair per step, and a stage launches step i+1 only after the activation of step i
has arrived (`--handoff overlap` posts it behind the next launch instead).  `--scaling weak` (also reported as a second
key at N > 1): every rank owns a whole 32-layer block.  The rank-0 line carries `roofline` (dominant kernel, HIP events),
`cpu_baseline` (the oracle timed on the host cores of the same box, bounded sample), `strided_prefill` (configs[1] and the
wider strides of Bench-P), `dense_prefix` (the unscored causal prefix, MFMA-bound) and `boundary_kernels` (gather / scatter / in-place compaction bandwidth).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes(H, Hq, D, T, q_len, n_state, e=2):
    """W_step of SURVEY.md §8d per layer-step, split by kernel."""
    kv = 2 * H * T * D * e                       # read K and V once
    qo = 2 * Hq * q_len * D * e                  # q in, o out
    new = 2 * H * q_len * D * e                  # append new k, v
    state = 2 * n_state * H * T * 4              # score rows read + write
    return dict(attn=kv + qo // 2 + new, score=state + qo // 2, total=kv + qo + new + state)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, budget, policy, seconds=10.0):
    """The oracle (reference-shaped CPU path: torch.cat append, fp32 softmax, topk, boolean-mask
    compaction — easykv/easykv.py:56-68, :287-333) on a bounded sample of the same workload: WHOLE decode tokens over all
    `--layers` layers at the full T (no extrapolation from a few layers).  Headline: fp32 state on <= 16 threads, the sweet
    spot of these small torch ops on the GPU box's EPYC host; `variants` adds the two other forms SURVEY.md §8d names —
    fp16 storage (K/V kept in fp16, widened for the step and narrowed again, what a CPU run of the reference's fp16
    configuration pays) and the reference's default thread count (all cores)."""
    from oracle import easykv_oracle as O
    H = Hq = args.heads
    L, D, T = args.layers, args.head_dim, budget + 1
    g = torch.Generator().manual_seed(1234)
    base_k = torch.randn(1, H, budget, D, generator=g).half()
    base_v = torch.randn(1, H, budget, D, generator=g).half()

    def fresh_states(dtype):
        states = []
        for l in range(L):
            st = O.LayerState(k=torch.roll(base_k, l, dims=2).to(dtype), v=torch.roll(base_v, l, dims=2).to(dtype))
            st.s, st.q, st.c = O.init_state_decoding((H,), budget)
            st.s += torch.rand(H, T, generator=g) * 1e-3
            st.q += st.s ** 2
            states.append(st)
        return states

    plan = O.StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)

    def run(threads, fp16_storage, secs):
        """-> (tokens/s, whole tokens completed, layer-steps timed, seconds).  Bounded: stops inside a token once `secs` is over
        (a configuration that cannot finish one token in the time box is priced from the layer-steps it did finish)."""
        torch.set_num_threads(threads)
        states = fresh_states(torch.float16 if fp16_storage else torch.float32)
        n_ls, t0, el = 0, time.perf_counter(), 0.0
        while el <= secs and n_ls < 64 * L:
            for st in states:
                q = torch.randn(1, Hq, 1, D, generator=g).half().float()
                k = torch.randn(1, H, 1, D, generator=g).half().float()
                v = torch.randn(1, H, 1, D, generator=g).half().float()
                if fp16_storage:
                    st.k, st.v = st.k.float(), st.v.float()
                O.layer_step(st, q, k, v, plan)
                if fp16_storage:
                    st.k, st.v = st.k.half(), st.v.half()
                n_ls += 1
                el = time.perf_counter() - t0
                if el > secs and n_ls % L != 0 and n_ls < L:      # not even one token inside the box: stop here
                    break
            if n_ls % L != 0:
                break
        return (n_ls / L) / el, n_ls // L, n_ls, el

    ncpu = min(16, os.cpu_count() or 1)
    v0, n0, ls0, e0 = run(ncpu, False, seconds)
    out = dict(value=v0, unit="tokens/s", cores=ncpu, kind="port", kind_detail=f"port (the oracle), {ncpu} threads — the all-core variant SURVEY.md §8d names is in `variants`", cpu=_cpu_model(), host_threads_available=os.cpu_count(),
               sample=f"{n0} whole decode tokens x {L} layers ({ls0} layer-steps, {e0:.1f} s) at full T={T}, H={H}, D={D}, fp32 state, "
                      f"{policy}, reference-shaped (torch.cat append, topk, boolean-mask compaction)")
    variants = []
    v1, n1, ls1, e1 = run(ncpu, True, seconds * 0.6)
    variants.append(dict(name="fp16_storage", value=v1, unit="tokens/s", cores=ncpu, sample=f"{ls1} layer-steps ({n1} whole tokens x {L} layers), {e1:.1f} s"))
    if (os.cpu_count() or 1) > ncpu:
        v2, n2, ls2, e2 = run(os.cpu_count(), False, seconds * 0.5)
        variants.append(dict(name="all_cores_fp32", value=v2, unit="tokens/s", cores=os.cpu_count(),
                             sample=f"{ls2} layer-steps in {e2:.1f} s, per-token = {L} x mean layer-step (the reference's default: torch uses "
                                    f"every core; these small ops do not scale past ~16 threads)"))
    out["variants"] = variants
    torch.set_num_threads(ncpu)
    return out


def device_copy_gbs(dev, nbytes=1 << 30, iters=8):
    """Measured device-to-device copy bandwidth (read + write bytes / time), the practical ceiling SURVEY.md §8d asks to be
    reported next to the 8 TB/s spec."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.zero_()
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(dev)
    return 2.0 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def device_read_gbs(dev, nbytes=1 << 30, iters=10):
    """Read-only counterpart: the fastest stock reduction found on this GPU (row-wise amax over 1 GiB of fp32, 4096 rows);
    torch.sum / torch.max over the flat tensor reach 3.7-4.0 TB/s, this one ~6.0 TB/s."""
    x = torch.ones(4096, nbytes // 4 // 4096, dtype=torch.float32, device=dev)
    for _ in range(3):
        x.amax(dim=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        x.amax(dim=1)
    e1.record()
    torch.cuda.synchronize(dev)
    return float(x.numel() * 4) * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def event_overhead_us(dev, reps=32):
    """What a HIP-event pair adds around ONE kernel launch: events around a one-element kernel (whose own run time is ~2 us).
    Informational: `roofline.achieved` uses the raw event durations (conservative); the rocprofv3 kernel trace under
    profiles/ shows the pure kernel duration, which is shorter by about this much."""
    x = torch.zeros(1, device=dev)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        x.add_(1.0)
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize(dev)
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[4:])
    return v[len(v) // 2]


def prewarm(step, seconds, dev):
    """Untimed pre-warm of a secondary figure: the same step for `seconds` of wall time.  A launch shape timed right after its first
    use runs at lower clocks for hundreds of milliseconds (DESIGN.md §3.5 'a measurement trap': configs[3] 903 us per step behind 8
    warm-up steps, 857-863 us behind >= 25 ms of them) — the headline run has had --prewarm-s since round 1."""
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)


def strided_prefill(args, dev, n_chunks=48, warm=8, S=4096, stride=8, mode="encoding", budget=0.5, streaming=False, shape=None, pmc=True, prewarm_s=0.25):
    """Secondary figures (never `value`): the chunk phase of a strided prefill (SURVEY.md §8d Bench-P).  Default = BASELINE.json
    configs[1]: S=4096, stride 8, budget 0.5, kv_policy roco; also run at stride 64 / 96 and at the configs[3] shape
    (S=9994, stride 96).  The cache oscillates idx <-> idx+stride, every chunk step attends the retained slots with `stride`
    queries per head, scores and evicts `stride` slots per (layer, head); all layers in one launch (pair)."""
    from easykv_amd import KVBank, StepPlan, geometry
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    if shape is not None:          # (layers, query heads, KV heads) of another BASELINE config
        L, Hq, H = shape
    bp, idx, r_idx = geometry(mode, S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(4321)
    rnd = lambda h, n: torch.randn(L, h, n, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    if streaming:                  # keys cached un-rotated, RoPE by slot index on every read (easykv/llama_patch.py:310-327)
        from easykv_amd.api import rope_tables
        bank.set_rope(*rope_tables(idx + stride + 64, D))
    bank.load_rows(rnd(H, idx), rnd(H, idx))          # state after the dense prefix and the fill-up chunks
    if not args.identity_layout:                      # steady state of the chunk phase: rows recycled in place for many steps
        bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    # distinct inputs per step: re-using one chunk would append the same eight key rows over and over, whose identical scores
    # pile up as exact ties in the selection keys (an artefact no real prompt produces)
    n_in = 2 * warm + n_chunks + 8
    qs_, ks_, vs_ = [rnd(Hq, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)]
    plan = StepPlan(policy=args.policy if args.policy in ("roco", "h2o_head", "tova") else "roco", phase="prefill", accumulate=True, evict=True,
                    budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, tova_head_mean=True, streaming=streaming)
    out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
    ids = torch.empty(L, H, stride, dtype=torch.int32, device=dev)
    # whole step as the library runs it (phases = 0: one launch when the scorer fuses into the attention kernel) ...
    # (one HIP-event pair around the timed region: a pair per step costs ~8 us of marker latency, see event_overhead_us)
    if prewarm_s > 0:      # clocks and score state settle on fresh inputs of the same distribution (new rows every step, like the timed ones)
        prewarm(lambda: bank.attend(plan, rnd(Hq, stride), rnd(H, stride), rnd(H, stride), out=out, evict_ids=ids), prewarm_s, dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(warm + n_chunks):
        if i == warm:
            ev[0].record()
        bank.attend(plan, qs_[i], ks_[i], vs_[i], out=out, evict_ids=ids)
    ev[1].record()
    torch.cuda.synchronize(dev)
    t_step = ev[0].elapsed_time(ev[1]) / n_chunks * 1e-3
    one_launch = bool(bank.step_plan(plan, stride)[1])     # what the library's own dispatch says (two passes = 3 launches)
    # ... and the same step as two launches (attention kernel, then fold + score + select + compaction), for the breakdown
    ev2 = []
    for i in range(warm + 8):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        q, k, v = qs_[warm + n_chunks + i], ks_[warm + n_chunks + i], vs_[warm + n_chunks + i]
        e[0].record()
        bank.attend(plan, q, k, v, out=out, evict_ids=ids, phases=1)     # chunk attention kernel
        e[1].record()
        bank.attend(plan, q, k, v, out=out, evict_ids=ids, phases=2)     # fold + score + select + compaction
        e[2].record()
        if i >= warm:
            ev2.append(e)
    torch.cuda.synchronize(dev)
    t_attn = sum(a.elapsed_time(b) for a, b, _ in ev2) / len(ev2) * 1e-3
    t_score = sum(b.elapsed_time(c) for _, b, c in ev2) / len(ev2) * 1e-3
    T = idx + stride
    n_state = {"roco": 3, "h2o_head": 1, "tova": 1}[plan.policy]
    by = algorithmic_bytes(H, Hq, D, T, stride, n_state)
    traffic, traffic_src = (None, None) if (streaming or shape is not None or not pmc) else prefill_pmc(S, stride, L, Hq, H, D, plan.policy)
    gbs = by["total"] * L / t_step / 1e9
    return {"workload": f"bench-P chunk phase: S={S} stride={stride} budget={budget:.4g} ({mode} geometry) -> idx={idx}, T={T}, L={L} Hq={Hq} H={H} D={D} "
                        f"kv_policy={plan.policy}" + (", streaming=True (RoPE by slot index on every read)" if streaming else ""),
            "value": stride / t_step, "unit": "prompt tokens/s (chunk phase, attention/eviction path only)",
            "us_per_chunk_step": t_step * 1e6, "one_launch": one_launch,
            "as_two_launches_us": {"attn_kernel": t_attn * 1e6, "score_select": t_score * 1e6},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "bytes_per_step": by["total"] * L, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": (traffic / (by["total"] * L)) if traffic else None,
                         "timing": "one HIP event pair around the timed chunk steps / steps (launches back to back)"},
            "chunk_steps_timed": n_chunks, "prewarm_s": prewarm_s, "steps_run": 2 * warm + n_chunks + 8, "slot_map": "identity" if args.identity_layout else "scattered"}


def prefill_pipeline(args, dev, rank, world, DS, S=9994, stride=96, n_chunks=32, warm=6):
    """N > 1 secondary figure: the chunk phase of BASELINE configs[3] (S=9994, stride 96, budget 0.5, roco: what the reference runs
    over 8 GPUs with device_map='auto', test_passkey.py:25-38) through the LAYER-SHARDED PIPELINE: rank r owns its LayerShard
    block of the --layers layers; chunk i's stage output [stride, Hq*D] fp16 goes r -> r+1 point to point (posted, not waited
    for: easykv_amd.dist.PipelineStage) and stage r starts chunk i+1 meanwhile — chunk i+1's input is the prompt, not chunk i's
    logits (easykv/easykv.py:426-433), and eviction state is per layer.  value = prompt tokens leaving the last stage per second
    (barrier + synchronize on both sides, max over ranks)."""
    from easykv_amd import KVBank, StepPlan, geometry
    Hq, D = args.heads, args.head_dim
    H = args.kv_heads or Hq
    shard = DS.LayerShard(rank, world, args.layers)
    Ls = shard.count
    k = seqs_per_launch(max(1, args.layers // world), H, args.seqs_per_launch, min_heads=512)      # in-flight sequences (prompts) whose chunk steps share a launch; one value for the job
    L = Ls * k
    bp, idx, _ = geometry("encoding", S, 0.5, stride)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    rnd = lambda h, n: torch.randn(L, h, n, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    n_in = warm + n_chunks
    qs_, ks_, vs_ = [rnd(Hq, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)]
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride)
    outs = [torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev) for _ in range(4)]     # (posted outputs stay alive)
    ids = torch.empty(L, H, stride, dtype=torch.int32, device=dev)
    stage = DS.PipelineStage(shard, depth=2)
    like = torch.zeros(k, stride, Hq * D, dtype=torch.float16, device=dev)
    t0 = 0.0
    for i in range(n_in):
        if i == warm:
            stage.drain()
            DS.barrier(dev)
            t0 = time.perf_counter()
        stage.recv_hidden(like)                    # the previous stage's output of THIS chunk (first stage: nothing to wait for)
        out = outs[i % 4]
        bank.attend(plan, qs_[i], ks_[i], vs_[i], out=out, evict_ids=ids)
        # posted; this stage carries on with chunk i+1 (the output of every sequence's last layer on this rank)
        stage.send_hidden(out.view(k, Ls, Hq, stride, D)[:, Ls - 1].transpose(1, 2).reshape(k, stride, Hq * D))
    stage.drain()
    DS.barrier(dev)
    dt = DS.max_over_ranks(time.perf_counter() - t0, dev)
    return {"workload": f"bench-P chunk phase through the layer pipeline: S={S} stride={stride} budget=0.5 (configs[3] shape), {args.layers} layers over "
                        f"{world} ranks ({Ls} on rank {rank}, {k} sequence(s) per launch), T={idx + stride}, Hq={Hq} H={H} D={D} roco",
            "value": k * n_chunks * stride / dt, "unit": "prompt tokens/s (chunk phase, attention/eviction path only, all stages)",
            "us_per_chunk_step_pipeline": dt / n_chunks * 1e6, "chunks_timed": n_chunks, "sequences_per_launch": k,
            "handoff": "isend of the stage output, up to 2 in flight; recv blocking", "max_outputs_in_flight_rank0": max(stage.run_ahead or [0])}


MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (the 2:1-sparsity figure is never used)


def dense_prefix(args, dev, S, stride, reps=3):
    """Secondary figure: the dense causal prefix of a strided prefill (reference easykv.py:396, :403-405 with keep_attention off:
    one forward over the first r_idx prompt tokens, no scoring).  All layers in one launch of the MFMA chunk kernel; flops =
    4 * Hq * D * r_idx^2 / 2 per layer (causal half of QK^T and PV)."""
    from easykv_amd import KVBank, StepPlan, geometry
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    _, _, n = geometry("encoding", S, 0.5, stride)
    g = torch.Generator(device=dev).manual_seed(99)
    q, k, v = (torch.randn(L, h, n, D, generator=g, device=dev).half() for h in (Hq, H, H))
    out = torch.empty(L, Hq, n, D, dtype=torch.float16, device=dev)
    plan = StepPlan(policy="full", phase="prefill", accumulate=False)
    ms = []
    warm_reps = 8                          # untimed: kernel load + clocks (a 9 ms launch timed cold reads ~5 % low)
    for _ in range(warm_reps + reps):
        bank = KVBank(L, Hq, H, D, cap=n + 8, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        bank.attend(plan, q, k, v, out=out)
        ev[1].record()
        torch.cuda.synchronize(dev)
        ms.append(ev[0].elapsed_time(ev[1]))
        del bank
    t = sum(ms[warm_reps:]) / reps * 1e-3
    fl = 4.0 * Hq * D * n * n / 2 * L
    return {"workload": f"dense causal prefix of S={S} stride={stride}: r_idx={n} tokens, L={L} Hq={Hq} H={H} D={D}, one launch",
            "ms": t * 1e3, "value": n / t, "unit": "prompt tokens/s (prefix, attention path only)",
            "roofline": {"bound": "mfma", "achieved": fl / t / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / t / 1e12 / MFMA_F16_PEAK_TFLOPS, "flops": fl, "traffic": None}}


def dense_prefix_scored(args, dev, n, kv_heads, stride, label, reps=3):
    """Secondary figure: the SCORED dense prefix of a strided prefill with keep_attention=True (reference easykv.py:396, :403-405,
    h2o_head_score :173-186: the prefix's probabilities seed S and Q): one step of ``n`` queries per layer — one pass for the
    output and the row statistics and a K-only column-sum pass of the wide-block kernel (the query blocks are walked inside the launch; the r x r map
    never exists) + the scorer.  flops = the attention's own 4 * Hq * D * n^2 / 2 per layer (causal QK^T and PV); the two-pass
    scheme executes 1.5x that on the MFMA pipe (QK^T twice)."""
    from easykv_amd import KVBank, StepPlan
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = kv_heads or Hq
    g = torch.Generator(device=dev).manual_seed(77)
    q, k, v = (torch.randn(L, h, n, D, generator=g, device=dev).half() for h in (Hq, H, H))
    out = torch.empty(L, Hq, n, D, dtype=torch.float16, device=dev)
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=False, stride=stride)
    ms = []
    warm_reps = 6
    for _ in range(warm_reps + reps):
        bank = KVBank(L, Hq, H, D, cap=n + stride + 8, device=dev)
        bank.state_init(n + stride, 1, stride)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        bank.attend(plan, q, k, v, out=out)
        ev[1].record()
        torch.cuda.synchronize(dev)
        ms.append(ev[0].elapsed_time(ev[1]))
        one_launch_set = bank.step_plan(plan, n)
        del bank
    t = sum(ms[warm_reps:]) / reps * 1e-3
    fl = 4.0 * Hq * D * n * n / 2 * L
    return {"workload": f"scored dense causal prefix ({label}): {n} tokens, L={L} Hq={Hq} H={H} D={D}, keep_attention, one pass + column-sum "
                        f"pass + scorer over all layers (n_split={one_launch_set[0]})",
            "ms": t * 1e3, "value": n / t, "unit": "prompt tokens/s (scored prefix, attention path only)",
            "roofline": {"bound": "mfma", "achieved": fl / t / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / t / 1e12 / MFMA_F16_PEAK_TFLOPS, "flops": fl, "mfma_flops_executed": 1.5 * fl, "traffic": None}}


def live_pmc_step(script_args, script, timeout_s=150, env=None):
    """HBM traffic of ONE chunk step whose work is several launches (one pass + column-sum pass + scorer), measured in THIS run:
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` in separate counter-only passes over a short child run of ``script``; the
    bytes of every launch of the path's kernels are summed and divided by the number of steps (= launches of the scorer, one per
    step).  gfx950 correction as in live_pmc: 2 x FETCH_SIZE + WRITE_SIZE, KiB.  -> (bytes per step, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    names = ("ekv_attn_wide_kernel", "ekv_attn_chunk_kernel", "ekv_score_select_kernel", "ekv_chunk_lds_kernel", "ekv_rope_q_kernel", "ekv_fold_kernel")
    tot, steps = {}, 0
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ekv_pmc_", dir="/tmp")
        try:
            # (the child says how many steps it ran: since round 5 a wide step has no scorer launch of its own to count them by)
            steps_file = os.path.join(d, "steps.txt")
            subprocess.run([rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, script] + script_args, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp", BENCH_CHUNK_STEPS_OUT=steps_file, **(env or {})), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == ctr and any(n in r["Kernel_Name"] for n in names)] if fs else []
            steps = int(open(steps_file).read()) if os.path.exists(steps_file) else 0
        except Exception as e:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"live PMC pass failed ({type(e).__name__})"
        shutil.rmtree(d, ignore_errors=True)
        if steps < 4:
            return None, f"live PMC pass saw {steps} steps"
        tot[ctr] = sum(float(r["Counter_Value"]) for r in rows) / steps
    return ((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0,
            f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/bench_chunk.py {' '.join(script_args)}"
            f"{(' [' + ' '.join(k + '=' + v for k, v in env.items()) + ']') if env else ''}, "
            "all launches of a step summed, 2 x FETCH + WRITE (gfx950 correction, KiB -> bytes)")


def prefill_pmc(S, stride, L, Hq, H, D, policy):
    """HBM bytes per whole chunk step from the newest rocprofv3 PMC summary under profiles/ (tools/prof_round.sh +
    tools/summarize_prof.py: FETCH_SIZE / WRITE_SIZE in separate passes, 2 x FETCH + WRITE): the kernels one step launches."""
    import glob
    import re
    stem = {(4096, 8): "c2", (4096, 64): "s64", (9994, 96): "c4"}.get((S, stride))
    if stem is None or (L, Hq, H, D, policy) != (32, 32, 32, 128, "roco"):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_prefill_summary.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    for f in reversed(files):
        try:
            ks = json.load(open(f))[stem]["kernels"]
        except Exception:
            continue
        one = [v for n, v in ks.items() if "ekv_chunk_lds_kernel" in n or ("ekv_attn_chunk_kernel" in n and "true>" in n)]
        if one:       # the whole step is one launch
            return one[0]["hbm_bytes_per_launch"], f"profiles/{os.path.basename(f)} [{stem}]: one launch per step"
        two = [v for n, v in ks.items() if "ekv_attn_chunk_kernel" in n or "ekv_attn_wide_kernel" in n or "ekv_score_select_kernel" in n]
        # steps of the profiled run = launches of the one pass (mode 0 instance `<.., 0>` of the wide-block kernel: once per step); a
        # summary from before round 5 counts them by the scorer launches (every step had one)
        steps = ([v["launches"] for n, v in ks.items() if "ekv_attn_wide_kernel" in n and n.rstrip().endswith(", 0>")] or
                 [v["launches"] for n, v in ks.items() if "ekv_score_select_kernel" in n])
        if two and steps:
            # launches per step from the launch counts: the two passes of the two-pass scheme may carry the same
            # kernel name (one template, two translation units), so that entry is the mean of the two and counts twice per step
            return (sum(v["hbm_bytes_per_launch"] * v["launches"] / steps[0] for v in two),
                    f"profiles/{os.path.basename(f)} [{stem}]: attention kernel launch(es) + scorer kernel of one step")
    return None, None


def boundary_kernels(args, dev, iters=6):
    """Bandwidth of the kernels at the drop-in boundary (not on the per-token path): the ordered gather that hands the legacy
    ``past_key_values`` tuple back (ekv_gather_ordered), the import of ordered rows (ekv_scatter_rows) and the reference-shaped
    physical compaction (ekv_compact_inplace, easykv/easykv.py:56-82), at the Llama2-7B shape.  Charges:
    gather / scatter move every row once in and once out, 2 x (2 H T D e) per layer; the in-place compaction moves the rows
    behind each head's first victim, 4 * sum_h (T - 1 - v_h) * D * e per layer (K and V, read + write)."""
    from easykv_amd import KVBank
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    T = args.budget + 1
    g = torch.Generator(device=dev).manual_seed(99)
    bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    k = torch.randn(L, H, T, D, generator=g, device=dev).half()
    v = torch.randn(L, H, T, D, generator=g, device=dev).half()
    bank.load_rows(k, v)
    perm = torch.argsort(torch.rand(L, H, T, generator=g, device=dev), dim=-1).int()

    def timed(fn, setup=None):
        ts = []
        for i in range(iters + 2):
            if setup is not None:
                setup()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize(dev)
            if i >= 2:
                ts.append(e0.elapsed_time(e1) * 1e-3)
        return sum(ts) / len(ts)

    out = {}
    io_bytes = 2 * (2 * L * H * T * D * 2)

    def scat():
        bank.n_slots = [0] * L
        bank.load_rows(k, v, pos_begin=0)
    t = timed(scat)
    out["ekv_scatter_rows"] = {"us": t * 1e6, "bytes": io_bytes, "gbs": io_bytes / t / 1e9, "layout": "identity"}
    t = timed(lambda: bank.ordered_kv())
    out["ekv_gather_ordered"] = {"us": t * 1e6, "bytes": io_bytes, "gbs": io_bytes / t / 1e9, "layout": "identity"}
    bank.slot_of_pos[:, :, :T] = perm
    t = timed(lambda: bank.ordered_kv())
    out["ekv_gather_ordered_scattered"] = {"us": t * 1e6, "bytes": io_bytes, "gbs": io_bytes / t / 1e9,
                                           "layout": "scattered slot map (random permutation of 256-byte rows)"}
    bank.reset()
    bank.load_rows(k, v)
    victims = torch.randint(0, T - 1, (L, H, 1), generator=g, device=dev, dtype=torch.int32)    # one victim per head (a decode step)
    moved = 4 * int((T - 1 - victims.long()).sum()) * D * 2

    def restore():
        bank.n_slots = [T] * L
    t = timed(lambda: bank.compact_inplace(victims), restore)
    out["ekv_compact_inplace"] = {"us": t * 1e6, "bytes": moved, "gbs": moved / t / 1e9, "victims_per_head": 1,
                                  "charge": "4 * sum_h (T - 1 - v_h) * D * e"}
    for name in out:
        out[name]["frac_of_hbm_peak"] = out[name]["gbs"] / HBM_PEAK_GBS
    out["shape"] = f"L={L} H={H} T={T} D={D} fp16"
    return out


def streaming_decode(args, dev, budget, policy):
    """Secondary figure: the Bench-D decode step with ``streaming=True`` (RoPE-on-read, easykv/llama_patch.py:310-327: keys cached
    un-rotated, rotated by their current position index on every read — fp32 tables, 512 table bytes per 256-byte key row from L2),
    all layers in one fused launch, same steady-state preparation as the headline run."""
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import rope_tables
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    T = budget + 1
    gen = torch.Generator(device=dev).manual_seed(4242)
    bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    bank.set_rope(*rope_tables(T + 128, D))
    bank.load_rows(torch.randn(L, H, budget, D, generator=gen, device=dev).half(), torch.randn(L, H, budget, D, generator=gen, device=dev).half())
    bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=gen, device=dev), dim=-1).int()
    bank.state_init(T, 0)
    n_in = 32
    qs = torch.randn(n_in, L, Hq, 1, D, generator=gen, device=dev).half()
    ks = torch.randn(n_in, L, H, 1, D, generator=gen, device=dev).half()
    vs = torch.randn(n_in, L, H, 1, D, generator=gen, device=dev).half()
    o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev)
    ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget, streaming=True)
    n_split, fused = bank.step_plan(plan, 1)
    t_end, i = time.perf_counter() + 0.3, 0
    while time.perf_counter() < t_end:
        for _ in range(32):
            bank.attend(plan, qs[i % n_in], ks[i % n_in], vs[i % n_in], out=o, evict_ids=ids)
            i += 1
        torch.cuda.synchronize(dev)
    n = 512
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for j in range(n):
        bank.attend(plan, qs[(i + j) % n_in], ks[(i + j) % n_in], vs[(i + j) % n_in], out=o, evict_ids=ids)
    ev[1].record()
    torch.cuda.synchronize(dev)
    t = ev[0].elapsed_time(ev[1]) / n * 1e-3
    b = algorithmic_bytes(H, Hq, D, T, 1, {"roco": 3, "h2o_head": 1, "tova": 1}.get(policy, 0))
    gbs = b["total"] * L / t / 1e9
    return {"workload": f"bench-D decode step with streaming=True (RoPE-on-read): L={L} Hq={Hq} H={H} D={D} T={T} {policy}", "us_per_step": t * 1e6,
            "value": 1.0 / t, "unit": "tokens/s", "plan": {"fused_one_launch": bool(fused), "n_split": n_split},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_step": b["total"] * L,
                         "note": "algorithmic bytes exclude the rotation tables (L2-resident)"}}


def seqs_per_launch(n_layers_of_rank, n_kv_heads, want=0, min_heads=256):
    """In-flight sequences a pipeline stage serves per launch.  A stage that owns few layers launches few heads (N = 8: 4 layers x 32
    = 128 heads, half a head per CU: the fused one-launch step needs >= 256, ekv_abi.hip) — but the pipeline holds >= N sequences in
    flight anyway, and the bank is generic in its layer count: (sequence, layer) pairs are just more layers.  Default:
    the fewest sequences (1, 2, 4 ...) that put >= ``min_heads`` heads into the launch: 256 for decode steps (one 8-wave workgroup
    per CU), 512 for wide chunk steps (two workgroups per CU with unsplit heads, whose scorer then runs as the tail of the
    column-sum pass: measured 0.31 of the HBM peak at 256 heads x 2 key-range splits)."""
    if want > 0:
        return want
    k = 1
    while k * n_layers_of_rank * n_kv_heads < min_heads and k < 8:
        k *= 2
    return k


def decode_config0(args, dev, P=37, budget=200, n=512):
    """BASELINE.json configs[0] at its own geometry (test_decoding.py:29-48: decoding mode, budget 200, roco; the reference runs it on
    the CPU in fp32): the decode step after the budget has filled — a prompt of P never-evicted tokens + W = 201 scored slots,
    recent window 60, k1 = 140 — all 32 layers in one launch.  A 4 MB-per-layer step: launch- and tail-bound, not a bandwidth figure."""
    from easykv_amd import KVBank, StepPlan
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    T = P + budget + 1
    gen = torch.Generator(device=dev).manual_seed(200)
    bank = KVBank(L, Hq, H, D, cap=T + 8, device=dev)
    bank.load_rows(torch.randn(L, H, P + budget, D, generator=gen, device=dev).half(), torch.randn(L, H, P + budget, D, generator=gen, device=dev).half())
    bank.state_init(budget + 1, 0)
    n_in = 32
    qs = torch.randn(n_in, L, Hq, 1, D, generator=gen, device=dev).half()
    ks = torch.randn(n_in, L, H, 1, D, generator=gen, device=dev).half()
    vs = torch.randn(n_in, L, H, 1, D, generator=gen, device=dev).half()
    o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev)
    ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy="roco", phase="decode", evict=True, score_off=P, budget=budget)
    n_split, fused = bank.step_plan(plan, 1)
    for i in range(256):
        bank.attend(plan, qs[i % n_in], ks[i % n_in], vs[i % n_in], out=o, evict_ids=ids)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(n):
        bank.attend(plan, qs[i % n_in], ks[i % n_in], vs[i % n_in], out=o, evict_ids=ids)
    ev[1].record()
    torch.cuda.synchronize(dev)
    t = ev[0].elapsed_time(ev[1]) / n * 1e-3
    by = 2 * H * T * D * 2 + 2 * Hq * D * 2 + 2 * H * D * 2 + 2 * 3 * H * (budget + 1) * 4      # W_step of §8d with the score rows over W = budget + 1
    return {"workload": f"configs[0] decode step: decoding mode, budget={budget}, prompt {P}, T={T}, W={budget + 1}, L={L} Hq={Hq} H={H} D={D} roco",
            "us_per_step": t * 1e6, "value": 1.0 / t, "unit": "tokens/s", "plan": {"fused_one_launch": bool(fused), "n_split": n_split},
            "roofline": {"bound": "hbm", "achieved": by * L / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by * L / t / 1e9 / HBM_PEAK_GBS, "bytes_per_step": by * L}}


def stage_workloads(args, dev, budget, policy):
    """Secondary figures: what ONE RANK of the layer-sharded model runs per step at N = 2 / 4 / 8 (strong scaling) —
    the Bench-D decode step with 16 / 8 / 4 of the 32 layers per sequence — measured on this one GPU so that the first real 1/2/4/8
    curve can be checked against a prediction: us per step, the library's plan (one fused launch or attention +
    scorer launches, key-range splits), the roofline fraction on the algorithmic bytes of the launch.  A stage with fewer than 256
    heads serves `sequences_per_launch` in-flight sequences per launch (seqs_per_launch above; round 5) — the single-sequence
    launch of the same stage is reported beside it.  Same steady-state preparation as the headline run (scattered slot map,
    pre-warmed score rows).  Plus the configs[3] chunk step of a 4-layer stage."""
    from easykv_amd import KVBank, StepPlan, geometry
    Hq, D = args.heads, args.head_dim
    H = args.kv_heads or Hq
    T = budget + 1
    n_state = {"roco": 3, "h2o_head": 1, "tova": 1}.get(policy, 0)
    b = algorithmic_bytes(H, Hq, D, T, 1, n_state)

    def decode_stage(Ls, k):
        L = Ls * k                                      # (sequence, layer) pairs in the launch
        gen = torch.Generator(device=dev).manual_seed(77 + L)
        bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
        bank.load_rows(torch.randn(L, H, budget, D, generator=gen, device=dev).half(), torch.randn(L, H, budget, D, generator=gen, device=dev).half())
        bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=gen, device=dev), dim=-1).int()
        bank.state_init(T, 0)
        n_in = 64
        qs = torch.randn(n_in, L, Hq, 1, D, generator=gen, device=dev).half()
        ks = torch.randn(n_in, L, H, 1, D, generator=gen, device=dev).half()
        vs = torch.randn(n_in, L, H, 1, D, generator=gen, device=dev).half()
        o = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev)
        ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
        plan = StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)
        if policy == "recency":
            plan.range_start = 0
        n_split, fused = bank.step_plan(plan, 1)
        t_end = time.perf_counter() + 0.25            # pre-warm: clocks + score state
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(32):
                bank.attend(plan, qs[i % n_in], ks[i % n_in], vs[i % n_in], out=o, evict_ids=ids)
                i += 1
            torch.cuda.synchronize(dev)
        n = 512
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for j in range(n):
            bank.attend(plan, qs[(i + j) % n_in], ks[(i + j) % n_in], vs[(i + j) % n_in], out=o, evict_ids=ids)
        ev[1].record()
        torch.cuda.synchronize(dev)
        del bank
        return ev[0].elapsed_time(ev[1]) / n * 1e-3, n_split, fused

    out = []
    for Ls in (16, 8, 4):
        if Ls >= args.layers:
            continue
        k = seqs_per_launch(Ls, H)
        t, n_split, fused = decode_stage(Ls, k)
        gbs = b["total"] * Ls * k / t / 1e9
        e = {"workload": f"decode step of a {Ls}-layer stage (one rank of N={args.layers // Ls}, strong scaling), {k} in-flight sequence(s) per launch: "
                         f"L={Ls} Hq={Hq} H={H} D={D} T={T} {policy}",
             "layers_in_launch": Ls * k, "layers_of_stage": Ls, "sequences_per_launch": k, "us_per_step": t * 1e6, "plan": {"fused_one_launch": bool(fused), "n_split": n_split},
             "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "bytes_per_step": b["total"] * Ls * k, "timing": "one HIP event pair around 512 back-to-back steps"},
             "predicted_pipeline_tokens_per_s": k / t}
        if k > 1:       # the same stage serving ONE sequence per launch (rounds 1-4)
            t1, ns1, fu1 = decode_stage(Ls, 1)
            e["single_sequence_launch"] = {"us_per_step": t1 * 1e6, "frac": b["total"] * Ls / t1 / 1e9 / HBM_PEAK_GBS, "plan": {"fused_one_launch": bool(fu1), "n_split": ns1},
                                           "predicted_pipeline_tokens_per_s": 1.0 / t1}
        out.append(e)
    # the configs[3] chunk step of a 4-layer stage (N = 8)
    S, stride, Ls = 9994, 96, 4
    if Ls < args.layers:
        bp, idx, _ = geometry("encoding", S, 0.5, stride)
        by = algorithmic_bytes(H, Hq, D, idx + stride, stride, 3)

        def chunk_stage(k):
            L = Ls * k
            gen = torch.Generator(device=dev).manual_seed(4321)
            rnd = lambda h, m: torch.randn(L, h, m, D, generator=gen, device=dev).half()
            bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
            bank.load_rows(rnd(H, idx), rnd(H, idx))
            bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=gen, device=dev), dim=-1).int()
            bank.state_init(idx + stride, 2, stride)
            plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride)
            ins = [(rnd(Hq, stride), rnd(H, stride), rnd(H, stride)) for _ in range(4)]
            o = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
            ids = torch.empty(L, H, stride, dtype=torch.int32, device=dev)
            n_split, fused = bank.step_plan(plan, stride)
            prewarm(lambda: bank.attend(plan, rnd(Hq, stride), rnd(H, stride), rnd(H, stride), out=o, evict_ids=ids), 0.2, dev)
            for j in range(8):
                bank.attend(plan, *ins[j % 4], out=o, evict_ids=ids)
            n = 48
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for j in range(n):
                bank.attend(plan, *ins[j % 4], out=o, evict_ids=ids)
            ev[1].record()
            torch.cuda.synchronize(dev)
            del bank
            return ev[0].elapsed_time(ev[1]) / n * 1e-3, n_split, fused

        k = seqs_per_launch(Ls, H, min_heads=512)
        t, n_split, fused = chunk_stage(k)
        gbs = by["total"] * Ls * k / t / 1e9
        e = {"workload": f"configs[3] chunk step of a 4-layer stage (one rank of N=8), {k} in-flight sequence(s) per launch: S={S} stride={stride} T={idx + stride} "
                         f"L={Ls} Hq={Hq} H={H} D={D} roco",
             "layers_in_launch": Ls * k, "layers_of_stage": Ls, "sequences_per_launch": k, "us_per_step": t * 1e6, "plan": {"fused_one_launch": bool(fused), "n_split": n_split},
             "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "bytes_per_step": by["total"] * Ls * k},
             "predicted_pipeline_prompt_tokens_per_s": k * stride / t}
        if k > 1:
            t1, ns1, fu1 = chunk_stage(1)
            e["single_sequence_launch"] = {"us_per_step": t1 * 1e6, "frac": by["total"] * Ls / t1 / 1e9 / HBM_PEAK_GBS, "plan": {"fused_one_launch": bool(fu1), "n_split": ns1}}
        out.append(e)
    return out


def per_layer_chunk_steps(args, dev, S, stride, n_steps=6, warm=3, mode="encoding", budget=0.5, streaming=False, shape=None):
    """Secondary figure: a chunk step of the strided prefill issued ONE LAYER PER CALL, as a decoder stack does (layer l + 1's queries
    depend on layer l's output): per layer the attention launches + fold, and — round 4, ekv_step.defer_layers for chunk steps — the
    scorers of all layers in ONE launch at the end of the forward; ``immediate`` is the same step with every layer's scorer on the
    critical path (rounds 1-3).  us per layer = wall time of a whole forward's calls / layers."""
    from easykv_amd import KVBank, StepPlan, geometry
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    if shape is not None:
        L, Hq, H = shape
    bp, idx, _ = geometry(mode, S, budget, stride)
    g = torch.Generator(device=dev).manual_seed(99)
    rnd = lambda h, n: torch.randn(L, h, n, D, generator=g, device=dev).half()
    res = {}
    for name, defer in (("deferred_scorer", True), ("immediate", False)):
        bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
        if streaming:
            from easykv_amd.api import rope_tables
            bank.set_rope(*rope_tables(idx + stride + 64, D))
        bank.load_rows(rnd(H, idx), rnd(H, idx))
        bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
        bank.state_init(idx + stride, 2, stride)
        plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, streaming=streaming)
        ins = [(rnd(Hq, stride), rnd(H, stride), rnd(H, stride)) for _ in range(2)]
        out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
        views = [[(q[l:l + 1], k[l:l + 1], v[l:l + 1], out[l:l + 1]) for l in range(L)] for (q, k, v) in ins]
        t0 = 0.0

        def forward(i):
            for l in range(L):
                q1, k1, v1, o1 = views[i % 2][l]
                bank.attend(plan, q1, k1, v1, layer_begin=l, out=o1, defer=defer)
            if defer:
                bank.flush()
        t_end = time.perf_counter() + 0.15      # pre-warm (clocks): whole forwards
        while time.perf_counter() < t_end:
            forward(0)
            forward(1)
            torch.cuda.synchronize(dev)
        for i in range(warm + n_steps):
            if i == warm:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            forward(i)
        torch.cuda.synchronize(dev)
        res[name] = (time.perf_counter() - t0) / n_steps / L * 1e6
        del bank
    by = algorithmic_bytes(H, Hq, D, idx + stride, stride, 3)
    us = res["deferred_scorer"]
    return {"workload": f"chunk step one layer per call: S={S} stride={stride} T={idx + stride} L={L} Hq={Hq} H={H} D={D} roco" + (", streaming=True" if streaming else ""),
            "us_per_layer": us, "us_per_layer_immediate_scorer": res["immediate"], "value": stride / (us * L * 1e-6),
            "unit": "prompt tokens/s (chunk phase, attention/eviction path only)",
            "roofline_step": {"bound": "hbm (launch- / latency-bound in practice: 32 heads per launch)", "achieved": by["total"] / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": by["total"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bytes_per_layer_step": by["total"],
                              "timing": "host wall clock over whole forwards / layers"}}


def decode_run(args, dev, rank, world, scaling, DS, want_seq):
    """One timed Bench-D run.  scaling 'strong': the `--layers`-layer model is split over the ranks (this rank owns its
    LayerShard block); 'weak': every rank owns `--layers` layers."""
    from easykv_amd import KVBank, StepPlan
    Hq, D, budget = args.heads, args.head_dim, args.budget
    H = args.kv_heads or Hq
    T = budget + 1
    shard = DS.LayerShard(rank, world, args.layers if scaling == "strong" else args.layers * world)
    Ls = shard.count                    # layers of this rank
    # in-flight sequences this stage serves per launch (seqs_per_launch): 1 unless the stage launches < 256 heads (N = 8)
    # (the HEADLINE run serves ONE sequence per launch at every N unless --seqs-per-launch says otherwise, so that `value`
    #  is a single-sequence figure comparable across rounds and across N; the k-sequence launches of a short stage are reported by the
    #  boundary-stage entries, each next to its own single-sequence figure)
    k = max(1, args.seqs_per_launch)      # (one value for the whole job)
    L = Ls * k                          # (sequence, layer) pairs of this rank's bank: pair s * Ls + l
    n_total = args.steps + args.warmup
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    # cache pre-filled to `budget` retained slots (synthetic warm state, SURVEY.md §8d Bench-D)
    for l0 in range(0, L, 8):
        lc = min(8, L - l0)
        bank.load_rows(torch.randn(lc, H, budget, D, generator=gen, device=dev).half(),
                       torch.randn(lc, H, budget, D, generator=gen, device=dev).half(), pos_begin=0, layer_begin=l0)
    if not args.identity_layout:
        # Long-run steady state: a score-driven policy recycles rows in place, so after a few thousand steps the birth order
        # of the live rows is a random permutation of their addresses.  Start there instead of at the (sequential) identity
        # layout a fresh bank has, so `--warmup` does not decide what is measured.
        perm = torch.argsort(torch.rand(L, H, budget, generator=gen, device=dev), dim=-1).int()
        bank.slot_of_pos[:, :, :budget] = perm
    bank.state_init(T, 0)
    qs = torch.randn(n_total, L, Hq, 1, D, generator=gen, device=dev).half()
    ks = torch.randn(n_total, L, H, 1, D, generator=gen, device=dev).half()
    vs = torch.randn(n_total, L, H, 1, D, generator=gen, device=dev).half()
    # two output buffers in turn: the stage output of step i (the attention output of the rank's LAST layer) is sent straight
    # from outs[i % 2] while step i + 1 writes the other one
    outs = [torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev) for _ in range(2)]
    hidden_in = [torch.zeros(k, Hq * D, dtype=torch.float16, device=dev) for _ in range(2)]
    hidden_out = [torch.zeros(k, Hq * D, dtype=torch.float16, device=dev) for _ in range(2)]
    ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy=args.policy, phase="decode", evict=True, score_off=0, budget=budget, n_split=args.n_split)
    if args.policy == "recency":
        plan.range_start = 0
    lpl = min(args.layers_per_launch or L, L)
    handoff_on = world > 1 and not args.no_handoff and not args.graph
    sync_handoff = handoff_on and args.handoff == "sync"

    pending = []      # requests of the hand-off still in flight (world > 1)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    n_split, fused = bank.step_plan(plan, 1, 0, lpl)
    if args.split_kernels:
        fused = False
    st = {"i": 0}

    def step(i, timed_idx=None, handoff=True):
        out = outs[st["i"] & 1]
        if handoff and sync_handoff:
            # a pipeline stage consumes the previous stage's activation: the launch is ordered behind its arrival
            for req in pending:
                req.wait()
            pending.clear()
        for l0 in range(0, L, lpl):
            lc = min(lpl, L - l0)
            a = (plan, qs[i, l0:l0 + lc], ks[i, l0:l0 + lc], vs[i, l0:l0 + lc])
            kw = dict(layer_begin=l0, out=out[l0:l0 + lc], evict_ids=ids[l0:l0 + lc])
            if fused:        # the whole step is one launch
                if timed_idx is not None and l0 == 0:
                    ev[timed_idx][0].record()
                    bank.attend(*a, **kw)
                    ev[timed_idx][1].record()
                else:
                    bank.attend(*a, **kw)
            elif timed_idx is not None and l0 == 0 and not args.overlap_scorer:
                ev[timed_idx][0].record()
                bank.attend(*a, phases=1, **kw)
                ev[timed_idx][1].record()
                bank.attend(*a, phases=2, **kw)
                ev[timed_idx][2].record()
            elif args.split_kernels:
                bank.attend(*a, phases=1, **kw)
                bank.attend(*a, phases=2, **kw)
            else:
                bank.attend(*a, overlap_scorer=args.overlap_scorer, **kw)
        if args.overlap_scorer and args.graph:
            bank.join()        # a captured step must end with every forked stream joined
        if handoff and handoff_on:   # pipeline hand-off of the stage output (north star)
            # the stage output of every in-flight sequence: the attention output of its LAST layer on this rank
            send = out[L - 1].view(1, Hq * D) if k == 1 else hidden_out[st["i"] & 1].copy_(out.view(k, Ls, Hq * D)[:, Ls - 1])
            if sync_handoff:
                pending[:] = DS.ring_handoff_async(send, hidden_in[st["i"] & 1], shard, None)
            else:   # posted after this step's kernels, waited for after the next launch: the transfer overlaps it
                pending[:] = DS.ring_handoff_async(send, hidden_in[st["i"] & 1], shard, pending)
        st["i"] += 1

    # Clock / state pre-warm (untimed, before the W warmup steps): the same step for --prewarm-s seconds of wall time.  A cold
    # GPU needs tens of ms of load before its clocks settle, and the roco state needs ~1000 steps to reach the steady state the
    # policy lives in (low-mean tokens outside the feasible set accumulate), so neither depends on how small W is.
    n_pre = 0
    if args.prewarm_s > 0:
        torch.cuda.synchronize()
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_s:
            for _ in range(16):
                step(n_pre % n_total, handoff=False)     # rank-local: the ranks run different numbers of pre-warm steps
                n_pre += 1
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)

    graph = None
    if args.graph:   # launch-bound regimes (per-layer launches): replay the step as one hipGraph
        sq, sk, sv = qs[0].clone(), ks[0].clone(), vs[0].clone()
        qs_src, ks_src, vs_src = qs, ks, vs
        qs, ks, vs = sq.unsqueeze(0), sk.unsqueeze(0), sv.unsqueeze(0)     # step() now reads the static inputs (index 0)
        st["i"] = 0
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step(0)
        torch.cuda.synchronize()

    # Kernel timing for the roofline.  Fused path (one kernel per step, launches back to back): ONE HIP-event pair around the
    # timed region, duration per launch = region / steps (an upper bound: it contains any gap between launches).  An event
    # pair around every launch costs ~6 us of marker latency per step, lowers `value` and still over-states the kernel time.
    # Split path (two kernels per step): per-step events, needed for the per-kernel breakdown.
    per_step_events = args.step_events or not fused or lpl != L or bool(args.graph)
    DS.barrier(dev)
    region = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    t0 = time.perf_counter()
    region[0].record()
    for i in range(args.steps):
        if graph is not None:
            sq.copy_(qs_src[args.warmup + i]); sk.copy_(ks_src[args.warmup + i]); sv.copy_(vs_src[args.warmup + i])
            graph.replay()
        else:
            step(args.warmup + i, i if per_step_events else None)
    region[1].record()
    for req in pending:       # the last hand-off belongs to the timed region
        req.wait()
    pending.clear()
    DS.barrier(dev)
    elapsed = DS.max_over_ranks(time.perf_counter() - t0, dev)
    slot_rows = bool(any(getattr(bank, "_slot_rows", [False])))      # layout of the score rows the timed steps ran on
    if graph is not None:   # per-kernel durations: a short eager pass with HIP events
        for i in range(args.steps):
            step(0, i)
        torch.cuda.synchronize()

    bank.join()
    torch.cuda.synchronize()
    assert all(n == budget for n in bank.n_slots), bank.n_slots

    # secondary figure (not `value`): the same step issued one layer per launch, as a real sequential model does
    seq = None
    if want_seq and lpl == L and not args.graph and L > 1:
        n_seq = max(4, min(16, args.steps))
        out = outs[0]
        # the per-layer views are made up front: a model hands over its own tensors, slicing is not part of the path
        views = [[(qs[i, l0:l0 + 1], ks[i, l0:l0 + 1], vs[i, l0:l0 + 1], out[l0:l0 + 1]) for l0 in range(L)] for i in range(n_seq)]
        for rep_ in range(2):     # first pass warms the code path
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(n_seq):
                for l0 in range(L):     # attention + fold per layer, ONE scorer launch per token (ekv_step.defer_layers)
                    q1, k1, v1, o1 = views[i][l0]
                    bank.attend(plan, q1, k1, v1, layer_begin=l0, out=o1, defer=True)
                bank.flush()
            torch.cuda.synchronize()
            seq = n_seq / (time.perf_counter() - ts)
    t_region = region[0].elapsed_time(region[1]) / args.steps * 1e-3
    return dict(elapsed=elapsed, t_region=t_region, ev=ev, per_step_events=per_step_events, n_split=n_split, fused=fused, lpl=lpl,
                L=L, Ls=Ls, k=k, rank_us=DS.all_gather_floats(t_region * 1e6, dev), shard=shard, n_pre=n_pre, seq=seq, handoff=handoff_on, sync_handoff=sync_handoff, T=T, H=H, slot_rows=slot_rows)


def latest_pmc_summary(L, Hq, H, D, budget, policy, lpl):
    """HBM traffic of the fused kernel from the newest rocprofv3 PMC summary committed under profiles/ (FETCH_SIZE / WRITE_SIZE in
    separate passes, 2 x FETCH + WRITE: the guide's gfx950 correction).  Counters cannot be read from inside this process;
    the summary is re-collected every round with the same command (tools/summarize_prof.py) and named per round."""
    import glob
    import re
    if (L, Hq, H, D, budget, policy, lpl) != (32, 32, 32, 128, 2048, "roco", 32):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_decode_summary.json")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    for f in reversed(files):
        try:
            pm = json.load(open(f)).get("pmc", {})
            k = [v for n, v in pm.items() if "ekv_decode_fused_kernel<128, 1, false" in n]
            if k and "hbm_bytes_per_launch" in k[0]:
                return k[0]["hbm_bytes_per_launch"], (f"profiles/{os.path.basename(f)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                                      "passes, 2 x FETCH + WRITE (gfx950 correction), same bench command")
        except Exception:
            continue
    return None, None


def live_pmc(extra_args, kernel_substr, timeout_s=150, script=None):
    """HBM traffic of the dominant kernel measured in THIS run: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate
    passes (counters only — never together with a trace), each over a short child run of this same bench command (16 timed steps),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (2 x FETCH_SIZE + WRITE_SIZE, KiB).  -> (bytes per launch, source) or
    (None, reason).  Same recipe as tools/prof_round.sh, which also keeps the raw files under profiles/."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ekv_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable]
        cmd += ([script] if script else [os.path.abspath(__file__), "--no-cpu-baseline", "--steps", "16", "--warmup", "4", "--prewarm-s", "0.05",
                                         "--no-prefill", "--no-boundary", "--no-live-pmc"]) + extra_args
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=False)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0]))
                 if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == ctr] if fs else []
        except Exception as e:      # a profiler hiccup must never cost the bench line
            shutil.rmtree(d, ignore_errors=True)
            return None, f"live PMC pass failed ({type(e).__name__})"
        shutil.rmtree(d, ignore_errors=True)
        if len(v) < 4:
            return None, f"live PMC pass saw {len(v)} launches of the kernel"
        v = v[2:]
        vals[ctr] = sum(v) / len(v)
    return ((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over 16 timed steps of this command, "
            "2 x FETCH + WRITE (gfx950 correction, KiB -> bytes)")


def respawn(args):
    """Re-execute this command under torch.distributed.run, one rank per GPU of this node (backend nccl = RCCL; `--same-device
    --backend gloo` puts every rank on cuda:0 for the 1-GPU test box).  Rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    if not args.same_device:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"[bench] --gpus {args.gpus} asked for, {n_dev} GPU(s) visible: refusing to run fewer ranks under that label "
                  "(--same-device --backend gloo runs every rank on cuda:0 for tests)", file=sys.stderr)
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / device-tensor sharing across processes needs it on this host driver
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=0)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--budget", type=int, default=2048)
    ap.add_argument("--policy", default="roco")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: 'strong' splits the ONE --layers-layer model over the ranks (--layers/N each, pipeline hand-off of the "
                         "real stage output); 'weak' gives every rank a whole --layers-layer block.  At N > 1 the other one is reported as a second key")
    ap.add_argument("--handoff", choices=("sync", "overlap"), default="sync",
                    help="sync: a stage launches step i+1 after the activation of step i has arrived (full pipeline, N sequences in "
                         "flight); overlap: the transfer is waited for after the next launch (2N sequences in flight)")
    ap.add_argument("--layers-per-launch", type=int, default=0, help="0 = all layers of the rank in one launch")
    ap.add_argument("--n-split", type=int, default=0, help="key-range splits per head (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-events", action="store_true", help="fused path: bracket every launch with its own HIP event pair instead "
                    "of one pair around the timed region (adds ~6 us of marker latency per step)")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from profiles/ instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--ordered-rows", action="store_true", help="keep the score rows in the ordered layout (A/B switch for the slot-indexed layout of ABI 6)")
    ap.add_argument("--no-prefill", action="store_true", help="skip the secondary strided-prefill (configs[1]) figures")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary-kernel bandwidth figures")
    ap.add_argument("--no-handoff", action="store_true")
    ap.add_argument("--no-second-scaling", action="store_true", help="N > 1: skip the run in the other scaling mode")
    ap.add_argument("--split-kernels", action="store_true", help="force the two-kernel path (attention + score/select)")
    ap.add_argument("--overlap-scorer", action="store_true", help="split path: run the scorer on side streams, off the critical path")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke tests)")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0 (multi-rank smoke test on a 1-GPU box)")
    ap.add_argument("--prewarm-s", type=float, default=1.0, help="untimed pre-warm of clocks and score state before the warmup steps (seconds)")
    ap.add_argument("--identity-layout", action="store_true", help="start from a fresh bank's identity slot map (position order == "
                    "address order) instead of the scattered steady-state layout")
    ap.add_argument("--graph", action="store_true", help="capture one step (all launches) in a hipGraph and replay it")
    ap.add_argument("--seqs-per-launch", type=int, default=0, help="in-flight sequences a rank serves per launch in the HEADLINE run (0 / 1 = one: `value` is always single-sequence tokens/s; the secondary stage entries use the fewest that put >= 256 heads "
                    "into the launch: 1 up to N = 4, 2 at N = 8 for the Llama2-7B shape)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (it used to run ONE rank
        # and label the line n_gpus = 1).  The reference's multi-GPU entry needs no launcher either (test_passkey.py:25-35).
        sys.exit(respawn(args))

    from easykv_amd import dist as DS
    if args.ordered_rows:
        from easykv_amd.engine import KVBank as _KVBank
        _KVBank.use_slot_rows = False
    if args.same_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = DS.init(args.backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world} in the environment; the launcher's WORLD_SIZE is what runs (n_gpus = {world})", file=sys.stderr)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1 and args.scaling == "strong" and args.layers < world:
        raise SystemExit("--scaling strong needs at least one layer per rank")

    Hq, D, budget = args.heads, args.head_dim, args.budget
    # who is there: one all-reduce of ones over the backend (RCCL when nccl) and the device every rank runs on
    ranks_seen = int(round(DS.sum_over_ranks(1.0, dev)))
    devices = [int(x) for x in DS.all_gather_floats(float(dev.index or 0), dev)]
    r = decode_run(args, dev, rank, world, args.scaling, DS, want_seq=(rank == 0 and world == 1))
    second = None
    if world > 1 and not args.no_second_scaling and not args.graph:
        other = "weak" if args.scaling == "strong" else "strong"
        r2 = decode_run(args, dev, rank, world, other, DS, want_seq=False)
        tokens2 = args.steps * (world if other == "weak" else 1)      # (single-sequence, like `value`)
        second = {"scaling": other, "value": tokens2 / r2["elapsed"], "unit": "tokens/s", "ms_per_step": r2["elapsed"] / args.steps * 1e3,
                  "layers_per_rank": r2["Ls"], "sequences_per_launch": r2["k"], "fused": r2["fused"], "n_split": r2["n_split"],
                  "note": "weak: every rank owns a whole 32-layer block (aggregate layer-parallel throughput)" if other == "weak" else
                          "strong: the 32-layer model split over the ranks"}
    pipe = None
    if world > 1 and not args.no_prefill and not args.graph and args.layers >= world:
        pipe = prefill_pipeline(args, dev, rank, world, DS)
    if rank == 0:
        L, H, T, lpl, fused, ev = r["L"], r["H"], r["T"], r["lpl"], r["fused"], r["ev"]
        n_state = {"roco": 3, "h2o_head": 1, "tova": 1}.get(args.policy, 0)
        b = algorithmic_bytes(H, Hq, D, T, 1, n_state)
        lc0 = min(lpl, L)
        t_region, per_step_events = r["t_region"], r["per_step_events"]
        t_attn = t_region if not per_step_events else (1.0 if args.overlap_scorer else sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps * 1e-3)
        Ls, kseq = r["Ls"], r["k"]
        # every step of a rank emits one token per in-flight sequence of its launch; strong: the pipeline's output is the last stage's
        # `value` = tokens/s of ONE sequence (strong) / of one sequence per rank (weak); with k sequences sharing every launch
        # (--seqs-per-launch k) the job's total is reported next to it as aggregate_tokens_per_s, never as `value`
        tokens = args.steps * (world if args.scaling == "weak" else 1)
        cfg = {"workload": f"bench-D decode at fixed budget: B=1 L={args.layers} Hq={Hq} H={H} D={D} budget={budget} "
                           f"T={T} kv_policy={args.policy} (Llama2-7B shape, budget=50% of S=4096)",
               "parallelism": (f"pp{world}: {args.layers} layers split into contiguous blocks, {Ls} per rank, point-to-point hand-off of the stage output"
                               + (f", {kseq} in-flight sequences per launch ({Ls * kseq} (sequence, layer) pairs: >= 256 heads for the one-launch step)" if kseq > 1 else "")
                               if args.scaling == "strong" else f"{world} x {Ls}-layer blocks, layer-parallel") if world > 1 else "1 GPU",
               "layers_per_launch": lpl, "layers_per_rank": Ls, "sequences_per_launch": kseq, "layer_block_of_rank0": [r["shard"].begin, r["shard"].end], "n_split": r["n_split"],
               "fused": fused, "slot_map": "identity" if args.identity_layout else "scattered (random permutation: long-run steady state)",
               "score_rows": "slot-indexed (ABI 6: S / Q rewritten per step, count base + birth once per row, no compaction)" if r["slot_rows"] else "ordered",

               "prewarm_steps": r["n_pre"], "hipgraph": bool(args.graph), "overlap_scorer": bool(args.overlap_scorer),
               "handoff": ("sync: launch ordered behind the previous stage's activation" if r["sync_handoff"] else "overlapped with the next launch") if r["handoff"] else False}
        line = {
            "metric": "decode_tokens_per_sec", "value": tokens / r["elapsed"], "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f16 storage / f32 accumulate",
            "data": "synthetic", "config": cfg}
        if kseq > 1:
            line["aggregate_tokens_per_s"] = tokens * kseq / r["elapsed"]
            line["single_sequence"] = {"value": tokens / r["elapsed"], "unit": "tokens/s", "note": f"every launch serves {kseq} in-flight sequences; `value` counts one of them"}
        if world > 1:
            line["ranks"] = {"backend": torch.distributed.get_backend(), "ranks_seen": ranks_seen, "devices": devices,
                             "us_per_step": [round(x, 2) for x in r["rank_us"]]}
            line["rccl_ranks_seen"] = ranks_seen if torch.distributed.get_backend() == "nccl" else 0
        if second is not None:
            line["second_scaling"] = second
        if pipe is not None:
            line["strided_prefill_pipeline"] = pipe
        if r["seq"] is not None:
            us_layer = 1e6 / r["seq"] / L      # wall time per layer call (attention + in-kernel fold) incl. its share of the deferred scorer
            line["per_layer_launches"] = {"value": r["seq"], "unit": "tokens/s", "us_per_layer": us_layer,
                                          "roofline_step": {"bound": "hbm (latency-bound in practice: one launch of 32 heads per layer)",
                                                            "achieved": b["total"] / (us_layer * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                            "frac": b["total"] / (us_layer * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                            "bytes_per_layer_step": b["total"], "timing": "host wall clock over whole tokens / layers"},
                                          "note": "same step issued one layer per call, as a sequential "
                                          "model does: attention + fold per layer, the scorers of all layers in one launch per token; "
                                          "latency-bound; not the headline value"}
        if fused:
            traffic, traffic_src = latest_pmc_summary(args.layers, Hq, H, D, budget, args.policy, lpl) if world == 1 else (None, None)
            if world == 1 and not args.no_live_pmc:
                passthrough = ["--layers", str(args.layers), "--heads", str(args.heads), "--kv-heads", str(args.kv_heads),
                               "--head-dim", str(args.head_dim), "--budget", str(args.budget), "--policy", args.policy]
                live, live_src = live_pmc(passthrough, "ekv_decode_fused_kernel")
                if live is not None:
                    traffic, traffic_src = live, live_src
                elif traffic is not None:
                    traffic_src += f" (live collection unavailable: {live_src})"
            gbs = b["total"] * lc0 / t_attn / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "ekv_decode_fused_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                "bytes_per_launch": b["total"] * lc0, "avg_launch_us": t_attn * 1e6}
            if world == 1:
                copy = device_copy_gbs(dev)
                line["roofline"]["device_copy_gbs"] = copy      # measured read+write copy bandwidth of this GPU
                line["roofline"]["frac_of_device_copy"] = gbs / copy
                rd = device_read_gbs(dev)
                line["roofline"]["device_read_gbs"] = rd        # best stock read-only kernel (torch row-wise amax) on this GPU
                line["roofline"]["frac_of_device_read"] = gbs / rd
                line["roofline"]["event_pair_around_1elem_kernel_us"] = event_overhead_us(dev)
            line["roofline"]["timing"] = ("HIP event pair around every launch" if per_step_events else
                                          "one HIP event pair around the timed region / steps (launches are back to back)")
        elif args.overlap_scorer:
            line["roofline"] = None     # kernels of different layers overlap: per-kernel event timing is not meaningful here
        else:
            t_score = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps * 1e-3
            attn_gbs = b["attn"] * lc0 / t_attn / 1e9
            step_gbs = b["total"] * lc0 / (t_attn + t_score) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "ekv_attn_decode_kernel", "achieved": attn_gbs, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": attn_gbs / HBM_PEAK_GBS, "traffic": None,
                                "bytes_per_launch": b["attn"] * lc0, "avg_launch_us": t_attn * 1e6}
            line["roofline_step"] = {"kernels": "ekv_attn_decode_kernel + ekv_decode_score_kernel", "achieved": step_gbs,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
                                     "bytes_per_step_launches": b["total"] * lc0, "avg_us": (t_attn + t_score) * 1e6,
                                     "score_select_us": t_score * 1e6}
        if world == 1 and not args.no_prefill and not args.graph and (args.layers, Hq, H, D) == (32, 32, 32, 128):
            line["stage_workloads"] = stage_workloads(args, dev, budget, args.policy)
            line["streaming_decode"] = streaming_decode(args, dev, budget, args.policy)
        if world == 1 and not args.no_prefill and not args.graph:
            line["strided_prefill"] = strided_prefill(args, dev)
            sp = line["strided_prefill"]
            if not args.no_live_pmc and sp.get("one_launch") and (args.layers, Hq, H, D, args.policy) == (32, 32, 32, 128, "roco"):
                # configs[1]: the whole chunk step is one launch of the logits-in-LDS kernel — its traffic measured in this run
                live, live_src = live_pmc(["4096", "8", "16"], "ekv_chunk_lds_kernel", script=os.path.join(ROOT, "tools", "bench_chunk.py"))
                if live is not None:
                    sp["roofline"].update(traffic=live, traffic_source=live_src.replace("of this command", "of tools/bench_chunk.py 4096 8"),
                                          traffic_over_algorithmic=live / sp["roofline"]["bytes_per_step"])
            line["strided_prefill_more"] = [strided_prefill(args, dev, S=4096, stride=64, n_chunks=24),
                                            strided_prefill(args, dev, S=4096, stride=96, n_chunks=16),
                                            strided_prefill(args, dev, S=9994, stride=96, n_chunks=16),
                                            # BASELINE configs[2]: Mistral GQA (8 KV heads), stride 16, budget 0.3
                                            strided_prefill(args, dev, S=4096, stride=16, n_chunks=24, budget=0.3, shape=(32, 32, 8)),
                                            # BASELINE configs[4]: Llama2-13B heads, ppl-mode geometry, streaming RoPE-on-read
                                            strided_prefill(args, dev, S=10253, stride=96, n_chunks=8, warm=4, mode="ppl", budget=4096 / 10253,
                                                            streaming=True, shape=(40, 40, 40))]
            if not args.no_live_pmc and (args.layers, Hq, H, D, args.policy) == (32, 32, 32, 128, "roco"):
                # wide strides: a step is several launches (one pass, column-sum pass, scorer) — all of them measured in this run
                more = line["strided_prefill_more"]
                for spm, sargs, env in ((more[0], ["4096", "64", "8"], None), (more[1], ["4096", "96", "8"], None), (more[2], ["9994", "96", "6"], None),
                                        (more[3], ["4096", "16", "8", "8"], {"BUDGET": "0.3"}),
                                        (more[4], ["10253", "96", "5"], {"MODE": "ppl", "BUDGET": repr(4096 / 10253), "STREAMING": "1", "SHAPE": "40,40,40"})):
                    live, live_src = live_pmc_step(sargs, os.path.join(ROOT, "tools", "bench_chunk.py"), env=env)
                    if live is not None:
                        spm["roofline"].update(traffic=live, traffic_source=live_src, traffic_over_algorithmic=live / spm["roofline"]["bytes_per_step"])
            line["per_layer_chunk_steps"] = [per_layer_chunk_steps(args, dev, 4096, 8), per_layer_chunk_steps(args, dev, 4096, 64),
                                             per_layer_chunk_steps(args, dev, 9994, 96),
                                             per_layer_chunk_steps(args, dev, 4096, 16, budget=0.3, shape=(32, 32, 8)),
                                             per_layer_chunk_steps(args, dev, 10253, 96, n_steps=4, mode="ppl", budget=4096 / 10253, streaming=True, shape=(40, 40, 40))]
            line["dense_prefix"] = [dense_prefix(args, dev, 4096, 8), dense_prefix(args, dev, 9994, 96)]
            # scored prefix (keep_attention): BASELINE configs[2] (Mistral GQA, stride 16, budget 0.3: r_idx = 1216) and a 4906-token MHA prefix
            line["dense_prefix_scored"] = [dense_prefix_scored(args, dev, 1216, 8, 16, "configs[2]: S=4096 stride=16 budget=0.3"),
                                           dense_prefix_scored(args, dev, 4906, 0, 96, "S=9994 stride=96 budget=0.5")]
        if world == 1 and not args.no_boundary and not args.graph:
            line["boundary_kernels"] = boundary_kernels(args, dev)
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args, budget, args.policy if args.policy in ("roco", "h2o_head", "tova") else "roco")
        if "strided_prefill_more" in line and (args.layers, Hq, H, D) == (32, 32, 32, 128):
            # every BASELINE config in the part of the line a truncated tail keeps: one short entry each (details above)
            c0 = decode_config0(args, dev)
            line["decode_config0"] = c0
            more, sp = line["strided_prefill_more"], line["strided_prefill"]
            short = lambda i, name, e, us: {"config": i, "workload": name, "us_per_step": round(us, 1), "frac": round(e["roofline"]["frac"], 4),
                                            "traffic_over_algorithmic": (round(e["roofline"]["traffic_over_algorithmic"], 3) if e["roofline"].get("traffic_over_algorithmic") else None)}
            line["configs"] = [short(0, "decode step, budget 200, roco, 32 layers per launch (4 MB per layer: launch-bound)", c0, c0["us_per_step"]),
                               short(1, "chunk step S=4096 stride 8 budget 0.5 roco (Llama2-7B shape)", sp, sp["us_per_chunk_step"]),
                               short(2, "chunk step S=4096 stride 16 budget 0.3 (Mistral GQA 8 KV heads)", more[3], more[3]["us_per_chunk_step"]),
                               short(3, "chunk step S=9994 stride 96 budget 0.5 roco (1 GPU: all 32 layers)", more[2], more[2]["us_per_chunk_step"]),
                               short(4, "chunk step S=10253 stride 96 ppl geometry budget 4096, streaming RoPE-on-read (Llama2-13B heads, 40 layers)", more[4], more[4]["us_per_chunk_step"])]
        # Key order of the ONE line: the contract keys first, the bulky secondary figures in the middle, and what a reader of a
        # truncated tail must still see LAST — cpu_baseline, strided_prefill (BASELINE configs[1]) and roofline (the
        # driver's stdout tail had lost configs[1]).
        tail_keys = [k for k in ("stage_workloads", "per_layer_launches", "cpu_baseline", "strided_prefill", "roofline_step", "roofline", "configs") if k in line]
        line = {**{k: v for k, v in line.items() if k not in tail_keys}, **{k: line[k] for k in tail_keys}}
        print(json.dumps(line))
    if world > 1:
        DS.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
