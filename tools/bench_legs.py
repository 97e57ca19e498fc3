"""Secondary legs of bench.py's rank-0 line (everything but the headline Bench-D decode run): the strided prefill of BASELINE.json's
other configs, the layer-sharded prefill pipeline, the dense prefix, boundary kernels, streaming decode, the per-stage
workloads of a
layer-sharded job and one-layer-per-call chunk steps.  Moved out of bench.py in round 6; bench.py imports them."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from tools.bench_common import (HBM_PEAK_GBS, MFMA_F16_PEAK_TFLOPS, ROOT, algorithmic_bytes, live_pmc, live_pmc_step,
    prewarm, seqs_per_launch)


def strided_prefill(args, dev, n_chunks=48, warm=8, S=4096, stride=8, mode="encoding", budget=0.5, streaming=False,
    shape=None, pmc=True, prewarm_s=0.25):
    """Secondary figures (never `value`): the chunk phase of a strided prefill (SURVEY.md §8d Bench-P).  Default = BASELINE.json
    configs[1]: S=4096, stride 8, budget 0.5, kv_policy roco; also run at stride 64 / 96 and at the configs[3] shape
    (S=9994, stride 96).  The cache oscillates idx <-> idx+stride, every chunk step attends the retained slots with
    `stride`
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
    # keys cached un-rotated, RoPE by slot index on every read (easykv/llama_patch.py:310-327)
    if streaming:
        from easykv_amd.api import rope_tables
        bank.set_rope(*rope_tables(idx + stride + 64, D))
    bank.load_rows(rnd(H, idx), rnd(H, idx))          # state after the dense prefix and the fill-up chunks
    # steady state of the chunk phase: rows recycled in place for many steps
    if not args.identity_layout:
        bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    # distinct inputs per step: re-using one chunk would append the same eight key rows over and over, whose identical
    # scores
    # pile up as exact ties in the selection keys (an artefact no real prompt produces)
    n_in = 2 * warm + n_chunks + 8
    qs_, ks_, vs_ = [rnd(Hq, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)], [rnd(H,
        stride) for _ in range(n_in)]
    plan = StepPlan(policy=args.policy if args.policy in ("roco", "h2o_head", "tova") else "roco", phase="prefill",
        accumulate=True, evict=True,
                    budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, tova_head_mean=True, streaming=streaming)
    out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
    ids = torch.empty(L, H, stride, dtype=torch.int32, device=dev)
    # whole step as the library runs it (phases = 0: one launch when the scorer fuses into the attention kernel) ...
    # (one HIP-event pair around the timed region: a pair per step costs ~8 us of marker latency, see event_overhead_us)
    # clocks and score state settle on fresh inputs of the same distribution (new rows every step, like the timed ones)
    if prewarm_s > 0:
        prewarm(lambda: bank.attend(plan, rnd(Hq, stride), rnd(H, stride), rnd(H, stride), out=out, evict_ids=ids),
            prewarm_s, dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(warm + n_chunks):
        if i == warm:
            ev[0].record()
        bank.attend(plan, qs_[i], ks_[i], vs_[i], out=out, evict_ids=ids)
    ev[1].record()
    torch.cuda.synchronize(dev)
    t_step = ev[0].elapsed_time(ev[1]) / n_chunks * 1e-3
    one_launch = bool(bank.step_plan(plan,
        stride)[1])     # what the library's own dispatch says (two passes = 3 launches)
    # A short step can outrun the Python loop that issues it (one ctypes call per step: ~40 us of host work): the same steps — same
    # launches, fresh inputs every step — replayed from ONE hipGraph, so that the figure is the GPU's.  Reported next to the eager
    # loop's; the step time of the line is the smaller of the two and says which.
    t_eager, timing = t_step, "one HIP event pair around the timed chunk steps / steps (launches back to back)"
    graph_steps = 0
    if t_step < 100e-6:
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(n_chunks):
                    bank.attend(plan, qs_[warm + i], ks_[warm + i], vs_[warm + i], out=out, evict_ids=ids)
            for _ in range(3):
                gr.replay()
            reps = 4
            ev[0].record()
            for _ in range(reps):
                gr.replay()
            ev[1].record()
            torch.cuda.synchronize(dev)
            t_graph = ev[0].elapsed_time(ev[1]) / (reps * n_chunks) * 1e-3
            graph_steps = (3 + reps) * n_chunks
            if t_graph < t_step:
                t_step = t_graph
                timing = (f"one HIP event pair around {reps} replays of a hipGraph of {n_chunks} consecutive chunk steps (the eager Python loop "
                          f"issues a step every {t_eager * 1e6:.1f} us: host-bound)")
        except Exception as e:      # (a step that cannot be captured keeps the eager figure)
            timing += f"; hipGraph capture failed: {type(e).__name__}"
    # ... and the same step as two launches (attention kernel, then fold + score + select + compaction), for the
    # breakdown
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
    traffic, traffic_src = (None, None) if (streaming or shape is not None or not pmc) else prefill_pmc(S, stride, L,
        Hq, H, D, plan.policy)
    gbs = by["total"] * L / t_step / 1e9
    return {"workload": f"bench-P chunk phase: S={S} stride={stride} budget={budget:.4g} ({mode} geometry) -> idx={idx}, T={T}, L={L} Hq={Hq} H={H} D={D} "
                        f"kv_policy={plan.policy}" + (", streaming=True (RoPE by slot index on every read)" if streaming else ""),
            "value": stride / t_step, "unit": "prompt tokens/s (chunk phase, attention/eviction path only)",
            "us_per_chunk_step": t_step * 1e6, "one_launch": one_launch,
            "as_two_launches_us": {"attn_kernel": t_attn * 1e6, "score_select": t_score * 1e6},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS,
                         "bytes_per_step": by["total"] * L, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": (traffic / (by["total"] * L)) if traffic else None,
                         "timing": timing},
            "us_per_chunk_step_eager_loop": t_eager * 1e6,
            "chunk_steps_timed": n_chunks, "prewarm_s": prewarm_s, "steps_run": 2 * warm + n_chunks + 8 + graph_steps,
                "slot_map": "identity" if args.identity_layout else "scattered"}


def prefill_pipeline(args, dev, rank, world, DS, S=9994, stride=96, n_chunks=32, warm=6):
    """N > 1 secondary figure: the chunk phase of BASELINE configs[3] (S=9994, stride 96, budget 0.5, roco: what the reference runs
    over 8 GPUs with device_map='auto', test_passkey.py:25-38) through the LAYER-SHARDED PIPELINE: rank r owns its
    LayerShard
    block of the --layers layers; chunk i's stage output [stride, Hq*D] fp16 goes r -> r+1 point to point (posted, not
    waited
    for: easykv_amd.dist.PipelineStage) and stage r starts chunk i+1 meanwhile — chunk i+1's input is the prompt, not
    chunk i's
    logits (easykv/easykv.py:426-433), and eviction state is per layer.  value = prompt tokens leaving the last stage
    per second
    (barrier + synchronize on both sides, max over ranks)."""
    from easykv_amd import KVBank, StepPlan, geometry
    Hq, D = args.heads, args.head_dim
    H = args.kv_heads or Hq
    shard = DS.LayerShard(rank, world, args.layers)
    Ls = shard.count
    # in-flight sequences (prompts) whose chunk steps share a launch; one value for the job
    k = seqs_per_launch(max(1, args.layers // world), H, args.seqs_per_launch, min_heads=512)
    L = Ls * k
    bp, idx, _ = geometry("encoding", S, 0.5, stride)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    rnd = lambda h, n: torch.randn(L, h, n, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    bank.load_rows(rnd(H, idx), rnd(H, idx))
    bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    n_in = warm + n_chunks
    qs_, ks_, vs_ = [rnd(Hq, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)], [rnd(H,
        stride) for _ in range(n_in)]
    plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1),
        sink=4, stride=stride)
    # (posted outputs stay alive)
    outs = [torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev) for _ in range(4)]
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
            "value": k * n_chunks * stride / dt,
                "unit": "prompt tokens/s (chunk phase, attention/eviction path only, all stages)",
            "us_per_chunk_step_pipeline": dt / n_chunks * 1e6, "chunks_timed": n_chunks, "sequences_per_launch": k,
            "handoff": "isend of the stage output, up to 2 in flight; recv blocking",
                "max_outputs_in_flight_rank0": max(stage.run_ahead or [0])}


def dense_prefix(args, dev, S, stride, reps=3):
    """Secondary figure: the dense causal prefix of a strided prefill (reference easykv.py:396, :403-405 with keep_attention off:
    one forward over the first r_idx prompt tokens, no scoring).  All layers in one launch of the MFMA chunk kernel;
    flops =
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
    h2o_head_score :173-186: the prefix's probabilities seed S and Q): one step of ``n`` queries per layer — one pass
    for the
    output and the row statistics and a K-only column-sum pass of the wide-block kernel (the query blocks are walked
    inside the launch; the r x r map
    never exists) + the scorer.  flops = the attention's own 4 * Hq * D * n^2 / 2 per layer (causal QK^T and PV); the
    two-pass
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
                         "frac": fl / t / 1e12 / MFMA_F16_PEAK_TFLOPS, "flops": fl, "mfma_flops_executed": 1.5 * fl,
                             "traffic": None}}


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
            entry = json.load(open(f))[stem]
            ks = entry["kernels"]
        except Exception:
            continue
        # chunk steps the profiled child ran, as the child itself reports them (tools/bench_chunk.py "steps_run", round 6; ADVICE r5:
        # counting them by kernel-name suffix double-counts run-time-rep instances and misses the RoPE logits instance)
        steps_said = int((entry.get("bench_chunk_line") or {}).get("steps_run") or 0)
        one = [v for n,
            v in ks.items() if "ekv_chunk_lds_kernel" in n or ("ekv_attn_chunk_kernel" in n and "true>" in n)]
        if one:       # the whole step is one launch
            return one[0]["hbm_bytes_per_launch"], f"profiles/{os.path.basename(f)} [{stem}]: one launch per step"
        two = [v for n,
            v in ks.items() if "ekv_attn_chunk_kernel" in n or "ekv_attn_wide_kernel" in n or "ekv_score_select_kernel" in n]
        # steps of the profiled run = launches of the one pass (mode 0 instance `<.., 0>` of the wide-block kernel: once
        # per step); a
        # summary from before round 5 counts them by the scorer launches (every step had one)
        steps = ([steps_said] if steps_said > 0 else
                 [v["launches"] for n, v in ks.items() if "ekv_attn_wide_kernel" in n and n.rstrip().endswith(", 0>")] or
                 [v["launches"] for n, v in ks.items() if "ekv_score_select_kernel" in n])
        if two and steps:
            # launches per step from the launch counts: the two passes of the two-pass scheme may carry the same
            # kernel name (one template, two translation units), so that entry is the mean of the two and counts twice
            # per step
            return (sum(v["hbm_bytes_per_launch"] * v["launches"] / steps[0] for v in two),
                    f"profiles/{os.path.basename(f)} [{stem}]: attention kernel launch(es) + scorer kernel of one step")
    return None, None


def boundary_kernels(args, dev, iters=6):
    """Bandwidth of the kernels at the drop-in boundary (not on the per-token path): the ordered gather that hands the legacy
    ``past_key_values`` tuple back (ekv_gather_ordered), the import of ordered rows (ekv_scatter_rows) and the
    reference-shaped
    physical compaction (ekv_compact_inplace, easykv/easykv.py:56-82), at the Llama2-7B shape.  Charges:
    gather / scatter move every row once in and once out, 2 x (2 H T D e) per layer; the in-place compaction moves the
    rows
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
    # one victim per head (a decode step)
    victims = torch.randint(0, T - 1, (L, H, 1), generator=g, device=dev, dtype=torch.int32)
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
    un-rotated, rotated by their current position index on every read — fp32 tables, 512 table bytes per 256-byte key
    row from L2),
    all layers in one fused launch, same steady-state preparation as the headline run."""
    from easykv_amd import KVBank, StepPlan
    from easykv_amd.api import rope_tables
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    T = budget + 1
    gen = torch.Generator(device=dev).manual_seed(4242)
    bank = KVBank(L, Hq, H, D, cap=T + 63, device=dev)
    bank.set_rope(*rope_tables(T + 128, D))
    bank.load_rows(torch.randn(L, H, budget, D, generator=gen, device=dev).half(), torch.randn(L, H, budget, D,
        generator=gen, device=dev).half())
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
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "bytes_per_step": b["total"] * L,
                         "note": "algorithmic bytes exclude the rotation tables (L2-resident)"}}


def decode_config0(args, dev, P=37, budget=200, n=512):
    """BASELINE.json configs[0] at its own geometry (test_decoding.py:29-48: decoding mode, budget 200, roco; the reference runs it on
    the CPU in fp32): the decode step after the budget has filled — a prompt of P never-evicted tokens + W = 201 scored
    slots,
    recent window 60, k1 = 140 — all 32 layers in one launch.  A 4 MB-per-layer step: launch- and tail-bound, not a bandwidth figure."""
    from easykv_amd import KVBank, StepPlan
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    T = P + budget + 1
    gen = torch.Generator(device=dev).manual_seed(200)
    bank = KVBank(L, Hq, H, D, cap=T + 8, device=dev)
    bank.load_rows(torch.randn(L, H, P + budget, D, generator=gen, device=dev).half(), torch.randn(L, H, P + budget,
        D, generator=gen, device=dev).half())
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
    # W_step of §8d with the score rows over W = budget + 1
    by = 2 * H * T * D * 2 + 2 * Hq * D * 2 + 2 * H * D * 2 + 2 * 3 * H * (budget + 1) * 4
    return {"workload": f"configs[0] decode step: decoding mode, budget={budget}, prompt {P}, T={T}, W={budget + 1}, L={L} Hq={Hq} H={H} D={D} roco",
            "us_per_step": t * 1e6, "value": 1.0 / t, "unit": "tokens/s", "plan": {"fused_one_launch": bool(fused),
                "n_split": n_split},
            "roofline": {"bound": "hbm", "achieved": by * L / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": by * L / t / 1e9 / HBM_PEAK_GBS, "bytes_per_step": by * L}}


def stage_workloads(args, dev, budget, policy):
    """Secondary figures: what ONE RANK of the layer-sharded model runs per step at N = 2 / 4 / 8 (strong scaling) —
    the Bench-D decode step with 16 / 8 / 4 of the 32 layers per sequence — measured on this one GPU so that the first
    real 1/2/4/8
    curve can be checked against a prediction: us per step, the library's plan (one fused launch or
    attention +
    scorer launches, key-range splits), the roofline fraction on the algorithmic bytes of the launch.  A stage with
    fewer than 256
    heads serves `sequences_per_launch` in-flight sequences per launch (seqs_per_launch above; round 5) — the
    single-sequence
    launch of the same stage is reported beside it.  Same steady-state preparation as the headline run (scattered slot
    map,
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
        bank.load_rows(torch.randn(L, H, budget, D, generator=gen, device=dev).half(), torch.randn(L, H, budget, D,
            generator=gen, device=dev).half())
        bank.slot_of_pos[:, :, :budget] = torch.argsort(torch.rand(L, H, budget, generator=gen, device=dev),
            dim=-1).int()
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
             "layers_in_launch": Ls * k, "layers_of_stage": Ls, "sequences_per_launch": k, "us_per_step": t * 1e6,
                 "plan": {"fused_one_launch": bool(fused), "n_split": n_split},
             "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": gbs / HBM_PEAK_GBS,
                          "bytes_per_step": b["total"] * Ls * k,
                              "timing": "one HIP event pair around 512 back-to-back steps"},
             "predicted_pipeline_tokens_per_s": k / t}
        if k > 1:       # the same stage serving ONE sequence per launch (rounds 1-4)
            t1, ns1, fu1 = decode_stage(Ls, 1)
            e["single_sequence_launch"] = {"us_per_step": t1 * 1e6, "frac": b["total"] * Ls / t1 / 1e9 / HBM_PEAK_GBS,
                "plan": {"fused_one_launch": bool(fu1), "n_split": ns1},
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
            plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp,
                recent=int(bp * 0.1), sink=4, stride=stride)
            ins = [(rnd(Hq, stride), rnd(H, stride), rnd(H, stride)) for _ in range(4)]
            o = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
            ids = torch.empty(L, H, stride, dtype=torch.int32, device=dev)
            n_split, fused = bank.step_plan(plan, stride)
            prewarm(lambda: bank.attend(plan, rnd(Hq, stride), rnd(H, stride), rnd(H, stride), out=o, evict_ids=ids),
                0.2, dev)
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
             "layers_in_launch": Ls * k, "layers_of_stage": Ls, "sequences_per_launch": k, "us_per_step": t * 1e6,
                 "plan": {"fused_one_launch": bool(fused), "n_split": n_split},
             "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": gbs / HBM_PEAK_GBS,
                          "bytes_per_step": by["total"] * Ls * k},
             "predicted_pipeline_prompt_tokens_per_s": k * stride / t}
        if k > 1:
            t1, ns1, fu1 = chunk_stage(1)
            e["single_sequence_launch"] = {"us_per_step": t1 * 1e6,
                "frac": by["total"] * Ls / t1 / 1e9 / HBM_PEAK_GBS, "plan": {"fused_one_launch": bool(fu1),
                    "n_split": ns1}}
        out.append(e)
    return out


def per_layer_chunk_steps(args, dev, S, stride, n_steps=6, warm=3, mode="encoding", budget=0.5, streaming=False,
    shape=None):
    """Secondary figure: a chunk step of the strided prefill issued ONE LAYER PER CALL, as a decoder stack does (layer l + 1's queries
    depend on layer l's output): per layer the attention launches + fold, and — round 4, ekv_step.defer_layers for chunk
    steps — the
    scorers of all layers in ONE launch at the end of the forward; ``immediate`` is the same step with every layer's
    scorer on the
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
        plan = StepPlan(policy="roco", phase="prefill", accumulate=True, evict=True, budget=bp, recent=int(bp * 0.1),
            sink=4, stride=stride, streaming=streaming)
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
        if defer:
            # the same forwards replayed from a hipGraph (what generation_config['hipgraph'] does with the steady-state chunk forward of a
            # real model, easykv_amd/api.py GraphedForward): the GPU's share of a layer step, without the 33 host calls per forward
            try:
                import copy
                plan_eager, plan = plan, copy.copy(plan)      # (a new plan object: the per-step state is rebuilt on the capture stream)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    forward(0)
                    forward(1)
                for _ in range(3):
                    gr.replay()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(n_steps):
                    gr.replay()
                ev[1].record()
                torch.cuda.synchronize(dev)
                res["hipgraph"] = ev[0].elapsed_time(ev[1]) * 1e3 / (2 * n_steps * L)
                plan = plan_eager
            except Exception as e:
                res["hipgraph"] = f"capture failed: {type(e).__name__}"
        del bank
    by = algorithmic_bytes(H, Hq, D, idx + stride, stride, 3)
    us = res["deferred_scorer"]
    return {"workload": f"chunk step one layer per call: S={S} stride={stride} T={idx + stride} L={L} Hq={Hq} H={H} D={D} roco" + (", streaming=True" if streaming else ""),
            "us_per_layer": us, "us_per_layer_immediate_scorer": res["immediate"], "us_per_layer_hipgraph": res.get("hipgraph"),
            "value": stride / (us * L * 1e-6),
            "unit": "prompt tokens/s (chunk phase, attention/eviction path only)",
            "roofline_step": {"bound": "hbm (launch- / latency-bound in practice: 32 heads per launch)",
                "achieved": by["total"] / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": by["total"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  "bytes_per_layer_step": by["total"],
                              "timing": "host wall clock over whole forwards / layers"}}
