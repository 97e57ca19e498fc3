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
key at N > 1): every rank owns a whole 32-layer block.  The rank-0 line carries `roofline` (dominant kernel, HIP
events),
`cpu_baseline` (the oracle timed on the host cores of the same box, bounded sample), `strided_prefill` (configs[1] and
the
wider strides of Bench-P), `dense_prefix` (the unscored causal prefix, MFMA-bound) and `boundary_kernels` (gather /
scatter / in-place compaction bandwidth).
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

# the secondary legs and the shared helpers live in tools/ (round 6: bench.py keeps the headline run, the CPU baseline
# and main())
from tools.bench_common import (HBM_PEAK_GBS, _cpu_model, algorithmic_bytes, device_copy_gbs, device_read_gbs,
    event_overhead_us,  # noqa: E402
                                latest_pmc_summary, live_pmc, live_pmc_step, prewarm, seqs_per_launch)
from tools.bench_legs import (boundary_kernels, decode_config0, dense_prefix, dense_prefix_scored,
    per_layer_chunk_steps,  # noqa: E402
                              prefill_pipeline, stage_workloads, streaming_decode, strided_prefill)


def cpu_baseline(args, budget, policy, seconds=10.0):
    """The oracle (reference-shaped CPU path: torch.cat append, fp32 softmax, topk, boolean-mask
    compaction — easykv/easykv.py:56-68, :287-333) on a bounded sample of the same workload: WHOLE decode tokens over
    all
    `--layers` layers at the full T (no extrapolation from a few layers).  Headline: fp32 state on <= 16 threads, the
    sweet
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
    out = dict(value=v0, unit="tokens/s", cores=ncpu, kind="port",
        kind_detail=f"port (the oracle), {ncpu} threads — the all-core variant SURVEY.md §8d names is in `variants`",
            cpu=_cpu_model(), host_threads_available=os.cpu_count(),
               sample=f"{n0} whole decode tokens x {L} layers ({ls0} layer-steps, {e0:.1f} s) at full T={T}, H={H}, D={D}, fp32 state, "
                      f"{policy}, reference-shaped (torch.cat append, topk, boolean-mask compaction)")
    variants = []
    v1, n1, ls1, e1 = run(ncpu, True, seconds * 0.6)
    variants.append(dict(name="fp16_storage", value=v1, unit="tokens/s", cores=ncpu,
        sample=f"{ls1} layer-steps ({n1} whole tokens x {L} layers), {e1:.1f} s"))
    if (os.cpu_count() or 1) > ncpu:
        v2, n2, ls2, e2 = run(os.cpu_count(), False, seconds * 0.5)
        variants.append(dict(name="all_cores_fp32", value=v2, unit="tokens/s", cores=os.cpu_count(),
                             sample=f"{ls2} layer-steps in {e2:.1f} s, per-token = {L} x mean layer-step (the reference's default: torch uses "
                                    f"every core; these small ops do not scale past ~16 threads)"))
    out["variants"] = variants
    torch.set_num_threads(ncpu)
    return out




def decode_run(args, dev, rank, world, scaling, DS, want_seq):
    """One timed Bench-D run.  scaling 'strong': the `--layers`-layer model is split over the ranks (this rank owns its
    LayerShard block); 'weak': every rank owns `--layers` layers."""
    from easykv_amd import KVBank, StepPlan
    Hq, D, budget = args.heads, args.head_dim, args.budget
    H = args.kv_heads or Hq
    T = budget + 1
    shard = DS.LayerShard(rank, world, args.layers if scaling == "strong" else args.layers * world)
    Ls = shard.count                    # layers of this rank
    # in-flight sequences this stage serves per launch (seqs_per_launch): 1 unless the stage launches < 256 heads (N =
    # 8)
    # (the HEADLINE run serves ONE sequence per launch at every N unless --seqs-per-launch says otherwise, so
    # that `value`
    # is a single-sequence figure comparable across rounds and across N; the k-sequence launches of a short stage are
    # reported by the
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
        # Long-run steady state: a score-driven policy recycles rows in place, so after a few thousand steps the birth
        # order
        # of the live rows is a random permutation of their addresses.  Start there instead of at the (sequential)
        # identity
        # layout a fresh bank has, so `--warmup` does not decide what is measured.
        perm = torch.argsort(torch.rand(L, H, budget, generator=gen, device=dev), dim=-1).int()
        bank.slot_of_pos[:, :, :budget] = perm
    bank.state_init(T, 0)
    qs = torch.randn(n_total, L, Hq, 1, D, generator=gen, device=dev).half()
    ks = torch.randn(n_total, L, H, 1, D, generator=gen, device=dev).half()
    vs = torch.randn(n_total, L, H, 1, D, generator=gen, device=dev).half()
    # two output buffers in turn: the stage output of step i (the attention output of the rank's LAST layer) is sent
    # straight
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
            send = out[L - 1].view(1, Hq * D) if k == 1 else hidden_out[st["i"] & 1].copy_(out.view(k, Ls, Hq * D)[:,
                Ls - 1])
            if sync_handoff:
                pending[:] = DS.ring_handoff_async(send, hidden_in[st["i"] & 1], shard, None)
            else:   # posted after this step's kernels, waited for after the next launch: the transfer overlaps it
                pending[:] = DS.ring_handoff_async(send, hidden_in[st["i"] & 1], shard, pending)
        st["i"] += 1

    # Clock / state pre-warm (untimed, before the W warmup steps): the same step for --prewarm-s seconds of wall time.
    # A cold
    # GPU needs tens of ms of load before its clocks settle, and the roco state needs ~1000 steps to reach the steady
    # state the
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
        # step() now reads the static inputs (index 0)
        qs, ks, vs = sq.unsqueeze(0), sk.unsqueeze(0), sv.unsqueeze(0)
        st["i"] = 0
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step(0)
        torch.cuda.synchronize()

    # Kernel timing for the roofline.  Fused path (one kernel per step, launches back to back): ONE HIP-event pair
    # around the
    # timed region, duration per launch = region / steps (an upper bound: it contains any gap between launches).  An
    # event
    # pair around every launch costs ~6 us of marker latency per step, lowers `value` and still over-states the kernel
    # time.
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
        views = [[(qs[i, l0:l0 + 1], ks[i, l0:l0 + 1], vs[i, l0:l0 + 1],
            out[l0:l0 + 1]) for l0 in range(L)] for i in range(n_seq)]
        for rep_ in range(2):     # first pass warms the code path
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(n_seq):
                # attention + fold per layer, ONE scorer launch per token (ekv_step.defer_layers)
                for l0 in range(L):
                    q1, k1, v1, o1 = views[i][l0]
                    bank.attend(plan, q1, k1, v1, layer_begin=l0, out=o1, defer=True)
                bank.flush()
            torch.cuda.synchronize()
            seq = n_seq / (time.perf_counter() - ts)
    t_region = region[0].elapsed_time(region[1]) / args.steps * 1e-3
    return dict(elapsed=elapsed, t_region=t_region, ev=ev, per_step_events=per_step_events, n_split=n_split,
        fused=fused, lpl=lpl,
                L=L, Ls=Ls, k=k, rank_us=DS.all_gather_floats(t_region * 1e6, dev), shard=shard, n_pre=n_pre, seq=seq,
                    handoff=handoff_on, sync_handoff=sync_handoff, T=T, H=H, slot_rows=slot_rows)


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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
        "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # dmabuf IPC: RCCL / device-tensor sharing across processes needs it on this host driver
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    ap.add_argument("--step-events", action="store_true",
        help="fused path: bracket every launch with its own HIP event pair instead "
                    "of one pair around the timed region (adds ~6 us of marker latency per step)")
    ap.add_argument("--no-live-pmc", action="store_true",
        help="take roofline.traffic from profiles/ instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--ordered-rows", action="store_true",
        help="keep the score rows in the ordered layout (A/B switch for the slot-indexed layout of ABI 6)")
    ap.add_argument("--no-prefill", action="store_true", help="skip the secondary strided-prefill (configs[1]) figures")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary-kernel bandwidth figures")
    ap.add_argument("--no-handoff", action="store_true")
    ap.add_argument("--no-second-scaling", action="store_true", help="N > 1: skip the run in the other scaling mode")
    ap.add_argument("--split-kernels", action="store_true", help="force the two-kernel path (attention + score/select)")
    ap.add_argument("--overlap-scorer", action="store_true",
        help="split path: run the scorer on side streams, off the critical path")
    ap.add_argument("--backend", default="nccl",
        help="torch.distributed backend (nccl = RCCL; gloo only for smoke tests)")
    ap.add_argument("--same-device", action="store_true",
        help="debug: every rank uses cuda:0 (multi-rank smoke test on a 1-GPU box)")
    ap.add_argument("--prewarm-s", type=float, default=1.0,
        help="untimed pre-warm of clocks and score state before the warmup steps (seconds)")
    ap.add_argument("--identity-layout", action="store_true",
        help="start from a fresh bank's identity slot map (position order == "
                    "address order) instead of the scattered steady-state layout")
    ap.add_argument("--graph", action="store_true", help="capture one step (all launches) in a hipGraph and replay it")
    ap.add_argument("--seqs-per-launch", type=int, default=0,
        help="in-flight sequences a rank serves per launch in the HEADLINE run (0 / 1 = one: `value` is always single-sequence tokens/s; the secondary stage entries use the fewest that put >= 256 heads "
                    "into the launch: 1 up to N = 4, 2 at N = 8 for the Llama2-7B shape)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (it used to run
        # ONE rank
        # and label the line n_gpus = 1).  The reference's multi-GPU entry needs no launcher either
        # (test_passkey.py:25-35).
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
        second = {"scaling": other, "value": tokens2 / r2["elapsed"], "unit": "tokens/s",
            "ms_per_step": r2["elapsed"] / args.steps * 1e3,
                  "layers_per_rank": r2["Ls"], "sequences_per_launch": r2["k"], "fused": r2["fused"],
                      "n_split": r2["n_split"],
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
        # every step of a rank emits one token per in-flight sequence of its launch; strong: the pipeline's output is
        # the last stage's
        # `value` = tokens/s of ONE sequence (strong) / of one sequence per rank (weak); with k sequences sharing every
        # launch
        # (--seqs-per-launch k) the job's total is reported next to it as aggregate_tokens_per_s, never as `value`
        tokens = args.steps * (world if args.scaling == "weak" else 1)
        cfg = {"workload": f"bench-D decode at fixed budget: B=1 L={args.layers} Hq={Hq} H={H} D={D} budget={budget} "
                           f"T={T} kv_policy={args.policy} (Llama2-7B shape, budget=50% of S=4096)",
               "parallelism": (f"pp{world}: {args.layers} layers split into contiguous blocks, {Ls} per rank, point-to-point hand-off of the stage output"
                               + (f", {kseq} in-flight sequences per launch ({Ls * kseq} (sequence, layer) pairs: >= 256 heads for the one-launch step)" if kseq > 1 else "")
                               if args.scaling == "strong" else f"{world} x {Ls}-layer blocks, layer-parallel") if world > 1 else "1 GPU",
               "layers_per_launch": lpl, "layers_per_rank": Ls, "sequences_per_launch": kseq,
                   "layer_block_of_rank0": [r["shard"].begin, r["shard"].end], "n_split": r["n_split"],
               "fused": fused,
                   "slot_map": "identity" if args.identity_layout else "scattered (random permutation: long-run steady state)",
               "score_rows": "slot-indexed (ABI 6: S / Q rewritten per step, count base + birth once per row, no compaction)" if r["slot_rows"] else "ordered",

               "prewarm_steps": r["n_pre"], "hipgraph": bool(args.graph), "overlap_scorer": bool(args.overlap_scorer),
               "handoff": ("sync: launch ordered behind the previous stage's activation" if r["sync_handoff"] else "overlapped with the next launch") if r["handoff"] else False}
        line = {
            "metric": "decode_tokens_per_sec", "value": tokens / r["elapsed"], "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                "dtype": "f16 storage / f32 accumulate",
            "data": "synthetic", "config": cfg}
        if kseq > 1:
            line["aggregate_tokens_per_s"] = tokens * kseq / r["elapsed"]
            line["single_sequence"] = {"value": tokens / r["elapsed"], "unit": "tokens/s",
                "note": f"every launch serves {kseq} in-flight sequences; `value` counts one of them"}
        if world > 1:
            line["ranks"] = {"backend": torch.distributed.get_backend(), "ranks_seen": ranks_seen, "devices": devices,
                             "us_per_step": [round(x, 2) for x in r["rank_us"]]}
            line["rccl_ranks_seen"] = ranks_seen if torch.distributed.get_backend() == "nccl" else 0
        if second is not None:
            line["second_scaling"] = second
        if pipe is not None:
            line["strided_prefill_pipeline"] = pipe
        if r["seq"] is not None:
            # wall time per layer call (attention + in-kernel fold) incl. its share of the deferred scorer
            us_layer = 1e6 / r["seq"] / L
            line["per_layer_launches"] = {"value": r["seq"], "unit": "tokens/s", "us_per_layer": us_layer,
                                          "roofline_step": {"bound": "hbm (latency-bound in practice: one launch of 32 heads per layer)",
                                                            "achieved": b["total"] / (us_layer * 1e-6) / 1e9,
                                                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                            "frac": b["total"] / (us_layer * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                            "bytes_per_layer_step": b["total"],
                                                                "timing": "host wall clock over whole tokens / layers"},
                                          "note": "same step issued one layer per call, as a sequential "
                                          "model does: attention + fold per layer, the scorers of all layers in one launch per token; "
                                          "latency-bound; not the headline value"}
        if fused:
            traffic, traffic_src = latest_pmc_summary(args.layers, Hq, H, D, budget, args.policy,
                lpl) if world == 1 else (None, None)
            if world == 1 and not args.no_live_pmc:
                passthrough = ["--layers", str(args.layers), "--heads", str(args.heads), "--kv-heads",
                    str(args.kv_heads),
                               "--head-dim", str(args.head_dim), "--budget", str(args.budget), "--policy", args.policy]
                live, live_src = live_pmc(passthrough, "ekv_decode_fused_kernel")
                if live is not None:
                    traffic, traffic_src = live, live_src
                elif traffic is not None:
                    traffic_src += f" (live collection unavailable: {live_src})"
            gbs = b["total"] * lc0 / t_attn / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "ekv_decode_fused_kernel", "achieved": gbs,
                "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                                    "traffic_source": traffic_src,
                                "bytes_per_launch": b["total"] * lc0, "avg_launch_us": t_attn * 1e6}
            if world == 1:
                copy = device_copy_gbs(dev)
                line["roofline"]["device_copy_gbs"] = copy      # measured read+write copy bandwidth of this GPU
                line["roofline"]["frac_of_device_copy"] = gbs / copy
                rd = device_read_gbs(dev)
                # best stock read-only kernel (torch row-wise amax) on this GPU
                line["roofline"]["device_read_gbs"] = rd
                line["roofline"]["frac_of_device_read"] = gbs / rd
                line["roofline"]["event_pair_around_1elem_kernel_us"] = event_overhead_us(dev)
            line["roofline"]["timing"] = ("HIP event pair around every launch" if per_step_events else
                                          "one HIP event pair around the timed region / steps (launches are back to back)")
        elif args.overlap_scorer:
            # kernels of different layers overlap: per-kernel event timing is not meaningful here
            line["roofline"] = None
        else:
            t_score = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps * 1e-3
            attn_gbs = b["attn"] * lc0 / t_attn / 1e9
            step_gbs = b["total"] * lc0 / (t_attn + t_score) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": "ekv_attn_decode_kernel", "achieved": attn_gbs,
                "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": attn_gbs / HBM_PEAK_GBS, "traffic": None,
                                "bytes_per_launch": b["attn"] * lc0, "avg_launch_us": t_attn * 1e6}
            line["roofline_step"] = {"kernels": "ekv_attn_decode_kernel + ekv_decode_score_kernel",
                "achieved": step_gbs,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
                                     "bytes_per_step_launches": b["total"] * lc0, "avg_us": (t_attn + t_score) * 1e6,
                                     "score_select_us": t_score * 1e6}
        if world == 1 and not args.no_prefill and not args.graph and (args.layers, Hq, H, D) == (32, 32, 32, 128):
            line["stage_workloads"] = stage_workloads(args, dev, budget, args.policy)
            line["streaming_decode"] = streaming_decode(args, dev, budget, args.policy)
        if world == 1 and not args.no_prefill and not args.graph:
            line["strided_prefill"] = strided_prefill(args, dev)
            sp = line["strided_prefill"]
            if not args.no_live_pmc and sp.get("one_launch") and (args.layers, Hq, H, D, args.policy) == (32, 32, 32,
                128, "roco"):
                # configs[1]: the whole chunk step is one launch of the logits-in-LDS kernel — its traffic measured in
                # this run
                live, live_src = live_pmc(["4096", "8", "16"], "ekv_chunk_lds_kernel", script=os.path.join(ROOT,
                    "tools", "bench_chunk.py"))
                if live is not None:
                    sp["roofline"].update(traffic=live, traffic_source=live_src.replace("of this command",
                        "of tools/bench_chunk.py 4096 8"),
                                          traffic_over_algorithmic=live / sp["roofline"]["bytes_per_step"])
            line["strided_prefill_more"] = [strided_prefill(args, dev, S=4096, stride=64, n_chunks=24),
                                            strided_prefill(args, dev, S=4096, stride=96, n_chunks=16),
                                            strided_prefill(args, dev, S=9994, stride=96, n_chunks=16),
                                            # BASELINE configs[2]: Mistral GQA (8 KV heads), stride 16, budget 0.3
                                            strided_prefill(args, dev, S=4096, stride=16, n_chunks=24, budget=0.3,
                                                shape=(32, 32, 8)),
                                            # BASELINE configs[4]: Llama2-13B heads, ppl-mode geometry, streaming
                                            # RoPE-on-read
                                            strided_prefill(args, dev, S=10253, stride=96, n_chunks=8, warm=4,
                                                mode="ppl", budget=4096 / 10253,
                                                            streaming=True, shape=(40, 40, 40)),
                                            # the Mistral shape at stride 8 and budget 0.5 (32 folded rows x 2064 keys): the LONG
                                            # shape of the logits-resident kernel (no PMC pass of its own)
                                            strided_prefill(args, dev, S=4096, stride=8, n_chunks=24, budget=0.5,
                                                shape=(32, 32, 8), pmc=False)]
            if not args.no_live_pmc and (args.layers, Hq, H, D, args.policy) == (32, 32, 32, 128, "roco"):
                # wide strides: a step is several launches (one pass, column-sum pass, scorer) — all of them measured in
                # this run
                more = line["strided_prefill_more"]
                for spm, sargs, env in ((more[0], ["4096", "64", "8"], None), (more[1], ["4096", "96", "8"], None),
                    (more[2], ["9994", "96", "6"], None),
                                        (more[3], ["4096", "16", "8", "8"], {"BUDGET": "0.3"}),
                                        (more[4], ["10253", "96", "5"], {"MODE": "ppl", "BUDGET": repr(4096 / 10253),
                                            "STREAMING": "1", "SHAPE": "40,40,40"})):
                    live, live_src = live_pmc_step(sargs, os.path.join(ROOT, "tools", "bench_chunk.py"), env=env)
                    if live is not None:
                        spm["roofline"].update(traffic=live, traffic_source=live_src,
                            traffic_over_algorithmic=live / spm["roofline"]["bytes_per_step"])
            line["per_layer_chunk_steps"] = [per_layer_chunk_steps(args, dev, 4096, 8), per_layer_chunk_steps(args,
                dev, 4096, 64),
                                             per_layer_chunk_steps(args, dev, 9994, 96),
                                             per_layer_chunk_steps(args, dev, 4096, 16, budget=0.3, shape=(32, 32, 8)),
                                             per_layer_chunk_steps(args, dev, 10253, 96, n_steps=4, mode="ppl",
                                                 budget=4096 / 10253, streaming=True, shape=(40, 40, 40))]
            line["dense_prefix"] = [dense_prefix(args, dev, 4096, 8), dense_prefix(args, dev, 9994, 96)]
            # scored prefix (keep_attention): BASELINE configs[2] (Mistral GQA, stride 16, budget 0.3: r_idx = 1216) and
            # a 4906-token MHA prefix
            line["dense_prefix_scored"] = [dense_prefix_scored(args, dev, 1216, 8, 16,
                "configs[2]: S=4096 stride=16 budget=0.3"),
                                           dense_prefix_scored(args, dev, 4906, 0, 96, "S=9994 stride=96 budget=0.5")]
        if world == 1 and not args.no_boundary and not args.graph:
            line["boundary_kernels"] = boundary_kernels(args, dev)
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args, budget, args.policy if args.policy in ("roco", "h2o_head",
                "tova") else "roco")
        if "strided_prefill_more" in line and (args.layers, Hq, H, D) == (32, 32, 32, 128):
            # every BASELINE config in the part of the line a truncated tail keeps: one short entry each (details above)
            c0 = decode_config0(args, dev)
            line["decode_config0"] = c0
            more, sp = line["strided_prefill_more"], line["strided_prefill"]
            short = lambda i, name, e, us: {"config": i, "workload": name, "us_per_step": round(us, 1),
                "frac": round(e["roofline"]["frac"], 4),
                                            "traffic_over_algorithmic": (round(e["roofline"]["traffic_over_algorithmic"], 3) if e["roofline"].get("traffic_over_algorithmic") else None)}
            line["configs"] = [short(0,
                "decode step, budget 200, roco, 32 layers per launch (4 MB per layer: launch-bound)", c0,
                    c0["us_per_step"]),
                               short(1, "chunk step S=4096 stride 8 budget 0.5 roco (Llama2-7B shape)", sp,
                                   sp["us_per_chunk_step"]),
                               short(2, "chunk step S=4096 stride 16 budget 0.3 (Mistral GQA 8 KV heads)", more[3],
                                   more[3]["us_per_chunk_step"]),
                               short(3, "chunk step S=9994 stride 96 budget 0.5 roco (1 GPU: all 32 layers)", more[2],
                                   more[2]["us_per_chunk_step"]),
                               short(4, "chunk step S=10253 stride 96 ppl geometry budget 4096, streaming RoPE-on-read (Llama2-13B heads, 40 layers)", more[4], more[4]["us_per_chunk_step"])]
        # Key order of the ONE line: the contract keys first, the bulky secondary figures in the middle, and what a
        # reader of a
        # truncated tail must still see LAST — cpu_baseline, strided_prefill (BASELINE configs[1]) and roofline (VERDICT
        # r3: the
        # driver's stdout tail had lost configs[1]).
        tail_keys = [k for k in ("stage_workloads", "per_layer_launches", "cpu_baseline", "strided_prefill",
            "roofline_step", "roofline", "configs") if k in line]
        line = {**{k: v for k, v in line.items() if k not in tail_keys}, **{k: line[k] for k in tail_keys}}
        print(json.dumps(line))
    if world > 1:
        DS.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
