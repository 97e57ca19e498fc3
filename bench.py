#!/usr/bin/env python
"""Bench-D: decode at fixed budget over the budgeted-KV attention path.

One "step" = one decode token of the path: for each of the L=32 layers of the Llama2-7B shape
(B=1, Hq=H=32, D=128) a fused HIP step — append the new K/V row, attention of the query over the
T = budget+1 = 2049 retained slots, score accumulation (roco: sum p, sum p^2, count), victim
selection and slot-map/score-row compaction — so the cache stays at `budget` slots.  Inputs are
synthetic N(0,1) fp16 (resident in HBM before the timed region), weights do not exist on this
path.  The metric is BASELINE.json's: decode tokens/s (path only) + HBM GB/s of the kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by torch.distributed.run, one rank per GPU.  Layer blocks are independent
units (eviction state is per (layer, head)), so every rank owns a 32-layer block and runs the
same step with no data-path collective (weak scaling); the pipeline hand-off of the north star
(one [1, hidden] fp16 activation per stage boundary) is issued as an RCCL ring send/recv per step, posted
after the step's kernels and waited for at the next step (stages of a pipeline work on different tokens).
The rank-0 line also carries `roofline` (dominant kernel, HIP events) and `cpu_baseline` (the
oracle timed on the host cores of the same box, bounded sample).
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


def cpu_baseline(args, budget, policy, seconds=12.0):
    """The oracle (reference-shaped CPU path: torch.cat append, fp32 softmax, topk, boolean-mask
    compaction — easykv/easykv.py:56-68, :287-333) on a bounded sample of the same workload."""
    from oracle import easykv_oracle as O
    H = Hq = args.heads
    D, T = args.head_dim, budget + 1
    # 8-16 threads is the sweet spot of these small torch ops on the GPU box's EPYC host (measured: 8/16 threads
    # ~28 ms per layer-step, 64 threads ~100 ms, 256 threads seconds); the reference's default (all cores) is slower
    ncpu = min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncpu)
    g = torch.Generator().manual_seed(1234)
    L = 2
    states = []
    for _ in range(L):
        st = O.LayerState(k=torch.randn(1, H, budget, D, generator=g).half().float(),
                          v=torch.randn(1, H, budget, D, generator=g).half().float())
        st.s, st.q, st.c = O.init_state_decoding((H,), budget)
        st.s += torch.rand(H, T, generator=g) * 1e-3
        st.q += st.s ** 2
        states.append(st)
    plan = O.StepPlan(policy=policy, phase="decode", evict=True, score_off=0, budget=budget)
    n_ls, t0 = 0, time.perf_counter()
    while True:
        for st in states:
            q = torch.randn(1, Hq, 1, D, generator=g).half().float()
            k = torch.randn(1, H, 1, D, generator=g).half().float()
            v = torch.randn(1, H, 1, D, generator=g).half().float()
            O.layer_step(st, q, k, v, plan)
            n_ls += 1
        el = time.perf_counter() - t0
        if el > seconds or n_ls >= 4000:
            break
    per_token = el / n_ls * args.layers
    return dict(value=1.0 / per_token, unit="tokens/s", cores=ncpu, kind="port",
                sample=f"{n_ls} layer-steps (L={L} layers x {n_ls // L} decode steps) at full T={T}, H={H}, D={D}, fp32, "
                       f"{policy}; per-token = {args.layers} x mean layer-step ({el / n_ls * 1e3:.2f} ms)")


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


def strided_prefill(args, dev, n_chunks=48, warm=8):
    """Secondary figure (never `value`): BASELINE.json configs[1] — the chunk phase of a strided prefill, S=4096, stride 8,
    budget 0.5, kv_policy roco (SURVEY.md §8d Bench-P): the cache oscillates idx <-> idx+stride, every chunk step attends
    the retained slots with 8 queries per head, scores and evicts 8 slots per (layer, head); all layers in one launch pair."""
    from easykv_amd import KVBank, StepPlan, geometry
    L, Hq, D = args.layers, args.heads, args.head_dim
    H = args.kv_heads or Hq
    S, stride = 4096, 8
    bp, idx, r_idx = geometry("encoding", S, 0.5, stride)
    g = torch.Generator(device=dev).manual_seed(4321)
    rnd = lambda h, n: torch.randn(L, h, n, D, generator=g, device=dev).half()
    bank = KVBank(L, Hq, H, D, cap=idx + stride, device=dev)
    bank.load_rows(rnd(H, idx), rnd(H, idx))          # state after the dense prefix and the fill-up chunks
    if not args.identity_layout:                      # steady state of the chunk phase: rows recycled in place for many steps
        bank.slot_of_pos[:, :, :idx] = torch.argsort(torch.rand(L, H, idx, generator=g, device=dev), dim=-1).int()
    bank.state_init(idx + stride, 2, stride)
    # distinct inputs per step: re-using one chunk would append the same eight key rows over and over, whose identical scores
    # pile up as exact ties in the selection keys (an artefact no real prompt produces)
    n_in = 2 * warm + n_chunks + 8
    qs_, ks_, vs_ = [rnd(Hq, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)], [rnd(H, stride) for _ in range(n_in)]
    plan = StepPlan(policy=args.policy if args.policy in ("roco", "h2o_head", "tova") else "roco", phase="prefill", accumulate=True, evict=True,
                    budget=bp, recent=int(bp * 0.1), sink=4, stride=stride, tova_head_mean=True)
    out = torch.empty(L, Hq, stride, D, dtype=torch.float16, device=dev)
    ids = torch.empty(L, H, stride, dtype=torch.int32, device=dev)
    # whole step as the library runs it (phases = 0: one launch when the scorer fuses into the attention kernel) ...
    # (one HIP-event pair around the timed region: a pair per step costs ~8 us of marker latency, see event_overhead_us)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(warm + n_chunks):
        if i == warm:
            ev[0].record()
        bank.attend(plan, qs_[i], ks_[i], vs_[i], out=out, evict_ids=ids)
    ev[1].record()
    torch.cuda.synchronize(dev)
    t_step = ev[0].elapsed_time(ev[1]) / n_chunks * 1e-3
    one_launch = bool(bank.step_plan(plan, stride)[0] == 1)
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
    return {"workload": f"bench-P chunk phase: S={S} stride={stride} budget=0.5 -> idx={idx}, T={T}, L={L} Hq={Hq} H={H} D={D} kv_policy={plan.policy}",
            "value": stride / t_step, "unit": "prompt tokens/s (chunk phase, attention/eviction path only)",
            "us_per_chunk_step": t_step * 1e6, "one_launch_fused_scorer": one_launch,
            "as_two_launches_us": {"attn_kernel": t_attn * 1e6, "score_select": t_score * 1e6},
            "algorithmic_bytes_per_step": by["total"] * L, "achieved_gbs": by["total"] * L / t_step / 1e9,
            "frac_of_hbm_peak": by["total"] * L / t_step / 1e9 / HBM_PEAK_GBS, "chunk_steps_timed": n_chunks,
            "slot_map": "identity" if args.identity_layout else "scattered"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=0)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--budget", type=int, default=2048)
    ap.add_argument("--policy", default="roco")
    ap.add_argument("--layers-per-launch", type=int, default=0, help="0 = all layers of the rank in one launch")
    ap.add_argument("--n-split", type=int, default=0, help="key-range splits per head (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-events", action="store_true", help="fused path: bracket every launch with its own HIP event pair instead "
                    "of one pair around the timed region (adds ~6 us of marker latency per step)")
    ap.add_argument("--no-prefill", action="store_true", help="skip the secondary strided-prefill (configs[1]) figure")
    ap.add_argument("--no-handoff", action="store_true")
    ap.add_argument("--split-kernels", action="store_true", help="force the two-kernel path (attention + score/select)")
    ap.add_argument("--overlap-scorer", action="store_true", help="split path: run the scorer on side streams, off the critical path")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke tests)")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0 (multi-rank smoke test on a 1-GPU box)")
    ap.add_argument("--prewarm-s", type=float, default=0.4, help="untimed pre-warm of clocks and score state before the warmup steps (seconds)")
    ap.add_argument("--identity-layout", action="store_true", help="start from a fresh bank's identity slot map (position order == "
                    "address order) instead of the scattered steady-state layout")
    ap.add_argument("--graph", action="store_true", help="capture one step (all launches) in a hipGraph and replay it")
    args = ap.parse_args()

    from easykv_amd import dist as DS
    if args.same_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = DS.init(args.backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    shard = DS.LayerShard(rank, world, args.layers * world)   # weak scaling: every rank owns a block of `layers` layers

    from easykv_amd import KVBank, StepPlan

    L, Hq, D, budget = args.layers, args.heads, args.head_dim, args.budget
    H = args.kv_heads or Hq
    T = budget + 1
    n_total = args.steps + args.warmup
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    bank = KVBank(L, Hq, H, D, cap=T + 63)
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
    out = torch.empty(L, Hq, 1, D, dtype=torch.float16, device=dev)
    ids = torch.empty(L, H, 1, dtype=torch.int32, device=dev)
    plan = StepPlan(policy=args.policy, phase="decode", evict=True, score_off=0, budget=budget, n_split=args.n_split)
    if args.policy == "recency":
        plan.range_start = 0
    lpl = args.layers_per_launch or L
    hidden = torch.zeros(1, Hq * D, dtype=torch.float16, device=dev)
    hidden_in = torch.zeros_like(hidden)

    pending = []      # requests of the hand-off still in flight (world > 1)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    n_split, fused = bank.step_plan(plan, 1, 0, min(lpl, L))
    if args.split_kernels:
        fused = False

    def step(i, timed_idx=None, handoff=True):
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
        if handoff and world > 1 and not args.no_handoff and not args.graph:   # pipeline hand-off of the stage output (north star):
            # posted after this step's kernels, waited for at the next step, so the 8 KB transfer overlaps the next launch
            pending[:] = DS.ring_handoff_async(hidden, hidden_in, shard, pending)

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
    if graph is not None:   # per-kernel durations: a short eager pass with HIP events
        for i in range(args.steps):
            step(0, i)
        torch.cuda.synchronize()

    bank.join()
    torch.cuda.synchronize()
    assert all(n == budget for n in bank.n_slots), bank.n_slots

    # secondary figure (not `value`): the same step issued one layer per launch, as a real sequential model does
    seq = None
    if rank == 0 and lpl == L and not args.graph and L > 1:
        n_seq = max(4, min(16, args.steps))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(n_seq):
            for l0 in range(L):
                bank.attend(plan, qs[i, l0:l0 + 1], ks[i, l0:l0 + 1], vs[i, l0:l0 + 1], layer_begin=l0, out=out[l0:l0 + 1], evict_ids=ids[l0:l0 + 1])
        torch.cuda.synchronize()
        seq = n_seq / (time.perf_counter() - ts)
    if rank == 0:
        n_state = {"roco": 3, "h2o_head": 1, "tova": 1}.get(args.policy, 0)
        b = algorithmic_bytes(H, Hq, D, T, 1, n_state)
        lc0 = min(lpl, L)
        t_region = region[0].elapsed_time(region[1]) / args.steps * 1e-3
        t_attn = t_region if not per_step_events else (1.0 if args.overlap_scorer else sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps * 1e-3)
        cfg = {"workload": f"bench-D decode at fixed budget: B=1 L={L} Hq={Hq} H={H} D={D} budget={budget} "
                           f"T={T} kv_policy={args.policy} (Llama2-7B shape, budget=50% of S=4096)",
               "layers_per_launch": lpl, "layers_per_rank": L, "layer_block_of_rank0": [shard.begin, shard.end], "n_split": n_split, "fused": fused, "slot_map": "identity" if args.identity_layout else "scattered (random permutation: long-run steady state)", "prewarm_steps": n_pre, "hipgraph": bool(args.graph), "overlap_scorer": bool(args.overlap_scorer),
               "handoff": (world > 1 and not args.no_handoff and not args.graph)}
        line = {
            "metric": "decode_tokens_per_sec", "value": world * args.steps / elapsed, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 storage / f32 accumulate",
            "data": "synthetic", "config": cfg}
        if seq is not None:
            line["per_layer_launches"] = {"value": seq, "unit": "tokens/s", "note": "same step, one layer per launch (split path: "
                                          "attention kernel + scorer kernel per layer), latency-bound; not the headline value"}
        traffic, traffic_src = None, None
        summ = os.path.join(ROOT, "profiles", "r01_decode_summary.json")
        if fused and os.path.exists(summ) and (L, Hq, H, D, budget, args.policy, lpl) == (32, 32, 32, 128, 2048, "roco", 32):
            try:
                pm = json.load(open(summ)).get("pmc", {})
                k = [v for n, v in pm.items() if "ekv_decode_fused_kernel<128, 1, false" in n]
                if k and "hbm_bytes_per_launch" in k[0]:
                    traffic, traffic_src = k[0]["hbm_bytes_per_launch"], "profiles/r01_decode_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 2 x FETCH + WRITE (gfx950 correction), same command"
            except Exception:
                pass
        if fused:
            gbs = b["total"] * lc0 / t_attn / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "ekv_decode_fused_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                "bytes_per_launch": b["total"] * lc0, "avg_launch_us": t_attn * 1e6}
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
            line["roofline_step"] = {"kernels": "ekv_attn_decode_kernel + ekv_score_select_kernel", "achieved": step_gbs,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
                                     "bytes_per_step_launches": b["total"] * lc0, "avg_us": (t_attn + t_score) * 1e6,
                                     "score_select_us": t_score * 1e6}
        if world == 1 and not args.no_prefill and not args.graph:
            line["strided_prefill"] = strided_prefill(args, dev)
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args, budget, args.policy if args.policy in ("roco", "h2o_head", "tova") else "roco")
        print(json.dumps(line))
    if world > 1:
        DS.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
