"""Helpers shared by bench.py (the headline Bench-D run) and tools/bench_legs.py (its secondary legs): the algorithmic-byte model of
SURVEY.md §8d, device bandwidth probes, the pre-warm loop and the rocprofv3 --pmc child runs behind `roofline.traffic`."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
# dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (the 2:1-sparsity figure is never used)
MFMA_F16_PEAK_TFLOPS = 2500.0


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
    use runs at lower clocks for hundreds of milliseconds (docs/TUNING.md §3.5 'a measurement trap': configs[3] 903 us per
    step behind 8
    warm-up steps, 857-863 us behind >= 25 ms of them) — the headline run has had --prewarm-s since round 1."""
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)


def live_pmc_step(script_args, script, timeout_s=150, env=None):
    """HBM traffic of ONE chunk step whose work is several launches (one pass + column-sum pass + scorer), measured in THIS run:
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` in separate counter-only passes over a short child run of
    ``script``; the
    bytes of every launch of the path's kernels are summed and divided by the number of steps (= launches of the scorer,
    one per
    step).  gfx950 correction as in live_pmc: 2 x FETCH_SIZE + WRITE_SIZE, KiB.  -> (bytes per step, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    names = ("ekv_attn_wide_kernel", "ekv_attn_chunk_kernel", "ekv_score_select_kernel", "ekv_chunk_lds_kernel",
        "ekv_rope_q_kernel", "ekv_fold_kernel")
    # a step that is ONE launch of this kernel (ekv_attn_resident.inc): bytes per launch of that kernel alone — the child's "as two
    # launches" breakdown steps run the two-pass kernels, which are not what a step costs
    one_launch = "ekv_attn_resident_kernel"
    tot, steps, whole = {}, 0, False
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ekv_pmc_", dir="/tmp")
        try:
            # (the child says how many steps it ran: since round 5 a wide step has no scorer launch of its own to count
            # them by)
            steps_file = os.path.join(d, "steps.txt")
            subprocess.run([rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                script] + script_args, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp", BENCH_CHUNK_STEPS_OUT=steps_file, **(env or {})),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == ctr and any(n in r["Kernel_Name"] for n in names + (one_launch,))] if fs else []
            steps = int(open(steps_file).read()) if os.path.exists(steps_file) else 0
            ones = [r for r in rows if one_launch in r["Kernel_Name"]]
            if ones:
                rows, steps, whole = ones, len({r.get("Dispatch_Id", i) for i, r in enumerate(ones)}), True
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
            + ("mean over the launches of ekv_attn_resident_kernel (one launch = one step), " if whole else "all launches of a step summed, ")
            + "2 x FETCH + WRITE (gfx950 correction, KiB -> bytes)")


def latest_pmc_summary(L, Hq, H, D, budget, policy, lpl):
    """HBM traffic of the fused kernel from the newest rocprofv3 PMC summary committed under profiles/ (FETCH_SIZE / WRITE_SIZE in
    separate passes, 2 x FETCH + WRITE: the guide's gfx950 correction).  Counters cannot be read from inside this
    process;
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
    passes (counters only — never together with a trace), each over a short child run of this same bench command (16
    timed steps),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (2 x FETCH_SIZE + WRITE_SIZE, KiB).  -> (bytes per launch,
    source) or
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
        cmd += ([script] if script else [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "16", "--warmup",
            "4", "--prewarm-s", "0.05",
                                         "--no-prefill", "--no-boundary", "--no-live-pmc"]) + extra_args
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                stderr=subprocess.DEVNULL,
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


def seqs_per_launch(n_layers_of_rank, n_kv_heads, want=0, min_heads=256):
    """In-flight sequences a pipeline stage serves per launch.  A stage that owns few layers launches few heads (N = 8: 4 layers x 32
    = 128 heads, half a head per CU: the fused one-launch step needs >= 256, ekv_abi.hip) — but the pipeline holds >= N
    sequences in
    flight anyway, and the bank is generic in its layer count: (sequence, layer) pairs are just more
    layers.  Default:
    the fewest sequences (1, 2, 4 ...) that put >= ``min_heads`` heads into the launch: 256 for decode steps (one 8-wave
    workgroup
    per CU), 512 for wide chunk steps (two workgroups per CU with unsplit heads, whose scorer then runs as the tail of
    the
    column-sum pass: measured 0.31 of the HBM peak at 256 heads x 2 key-range splits)."""
    if want > 0:
        return want
    k = 1
    while k * n_layers_of_rank * n_kv_heads < min_heads and k < 8:
        k *= 2
    return k
