#!/usr/bin/env python
"""Condense a tools/prof_round.sh run (gpurun_out/prof_<tag>/...) into profiles/<tag>_*.

    python tools/summarize_prof.py gpurun_out/prof_r02 r02

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default bench command, ekv_* kernels),
profiles/<tag>_decode_summary.json (fused decode kernel: trace + PMC) and profiles/<tag>_prefill_summary.json (per chunk
shape: kernel durations from the PMC runs' timestamps + PMC bytes).

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies
the 128-byte requests of a wide coalesced streaming read (16 B per lane) at 64 bytes — exactly half — so it is doubled;
WRITE_SIZE is taken as is; the two counters are collected in separate passes and never together with a trace."""
import csv
import glob
import json
import os
import sys


def pmc_per_kernel(d, key):
    files = sorted(glob.glob(os.path.join(d, "*counter_collection.csv")), key=os.path.getmtime)    # newest run of this directory
    per = {}
    if not files:
        return per
    for r in csv.DictReader(open(files[-1])):
        if "ekv_" in r["Kernel_Name"] and r["Counter_Name"] == key:
            e = per.setdefault(r["Kernel_Name"], dict(v=[], dur=[], lds=int(r["LDS_Block_Size"]), vgpr=int(r["VGPR_Count"]),
                                                      wg=int(r["Workgroup_Size"]), grid=int(r["Grid_Size"])))
            e["v"].append(float(r["Counter_Value"]))
            e["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def combine(src, stem, skip_first=2):
    f, w = pmc_per_kernel(os.path.join(src, stem + "_fetch"), "FETCH_SIZE"), pmc_per_kernel(os.path.join(src, stem + "_write"), "WRITE_SIZE")
    out = {}
    for k in f:
        fv, wv = f[k]["v"][skip_first:] or f[k]["v"], (w.get(k, {}).get("v") or [0.0])
        wv = wv[skip_first:] or wv
        fetch, write = sum(fv) / len(fv), sum(wv) / len(wv)
        dur = f[k]["dur"][skip_first:] or f[k]["dur"]
        out[k] = dict(launches=len(f[k]["v"]), FETCH_SIZE_KiB_avg=fetch, WRITE_SIZE_KiB_avg=write,
                      hbm_bytes_per_launch=(2.0 * fetch + write) * 1024.0,
                      correction="2 x FETCH_SIZE (gfx950 wide coalesced reads tallied at half) + WRITE_SIZE, KiB -> bytes",
                      avg_us_under_pmc=sum(dur) / len(dur), workgroup_size=f[k]["wg"], lds_bytes=f[k]["lds"], vgprs=f[k]["vgpr"])
    return out


def main(src, tag):
    os.makedirs("profiles", exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")), key=os.path.getmtime)
    kernels = {}
    if stats:
        rows = list(csv.DictReader(open(stats[-1])))
        with open(f"profiles/{tag}_kernel_stats.csv", "w") as fo:
            w = csv.writer(fo)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                if "ekv_" in r["Name"]:
                    w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
                    kernels[r["Name"]] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3,
                                              max_us=float(r["MaxNs"]) / 1e3)
    dec = dict(command="rocprofv3 --kernel-trace --stats -- python bench.py --steps 512 --no-cpu-baseline ; rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE "
                       "-- python bench.py --no-cpu-baseline --steps 16 --warmup 4 --prewarm-s 0.05 --no-prefill --no-boundary",
               kernels=kernels, pmc=combine(src, "decode"))
    ut = os.path.join(src, "bench_under_trace.log")      # the traced process's own bench line: its avg_launch_us is what the trace's average must agree with
    if os.path.exists(ut):
        lines = [ln for ln in open(ut) if ln.startswith("{")]
        if lines:
            open(f"profiles/{tag}_bench_line_under_trace.json", "w").write(lines[-1])
            dec["bench_line_of_the_traced_process"] = {k: json.loads(lines[-1])[k] for k in ("value", "ms_per_step", "steps", "warmup")}
            dec["bench_line_of_the_traced_process"]["roofline"] = json.loads(lines[-1])["roofline"]
    bl = os.path.join(src, "bench_line.json")
    if os.path.exists(bl) and os.path.getsize(bl):
        dec["bench_line_same_box"] = json.loads(open(bl).read().strip().splitlines()[-1])
    json.dump(dec, open(f"profiles/{tag}_decode_summary.json", "w"), indent=1)
    pre = {}
    algo = dict(c2=("S=4096 stride=8 budget=0.5 (configs[1]): T=2064", 35663872 * 32), s64=("S=4096 stride=64: T=2176", None),
                c4=("S=9994 stride=96 budget=0.5 (configs[3] shape): T=5098", None),
                c5=("S=10253 stride=96 ppl geometry, streaming RoPE-on-read, L=Hq=H=40 (configs[4] shape): T=4205", None),
                c3m=("S=4096 stride=16 budget=0.3, 8 KV heads x GQA 4 (configs[2] shape): T=1248", None))
    for stem, (desc, _) in algo.items():
        c = combine(src, "chunk_" + stem)
        if c:
            log = os.path.join(src, f"chunk_{stem}_fetch.log")
            line = None
            if os.path.exists(log):
                for ln in open(log):
                    if ln.startswith("{"):
                        line = json.loads(ln)
            pre[stem] = dict(shape=desc, command=f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python tools/bench_chunk.py ...", kernels=c, bench_chunk_line=line)
            if line:
                ab = None
                for k, v in c.items():
                    if "chunk" in k and v["launches"] >= 8:
                        ab = v
                pre[stem]["note"] = "algorithmic bytes per step = algorithmic_bytes_per_step of bench.py (SURVEY.md §8d W_step x 32 layers)"
    json.dump(pre, open(f"profiles/{tag}_prefill_summary.json", "w"), indent=1)
    sq = os.path.join(src, "sq_counters.txt")
    if os.path.exists(sq) and os.path.getsize(sq):
        with open(f"profiles/{tag}_sq_counters.txt", "w") as fo:
            fo.write("# rocprofv3 --pmc (counters only, two passes) over tools/bench_prefix.py and tools/bench_chunk.py: tools/sq_counters.sh\n")
            fo.write(open(sq).read())
    ts = os.path.join(src, "wide_tail_stamps.txt")
    if os.path.exists(ts) and os.path.getsize(ts):
        with open(f"profiles/{tag}_wide_tail_stamps.txt", "w") as fo:
            fo.write("# cycle stamps (s_memtime) per head of the scorer tail of the wide column-sum pass, -DEKV_TAIL_PROFILE build: tools/experiments/exp_widetail_prof.py\n")
            fo.write("".join(ln for ln in open(ts) if "amdgpu.ids" not in ln))
    rs = os.path.join(src, "resident_stamps.txt")
    if os.path.exists(rs) and os.path.getsize(rs):
        with open(f"profiles/{tag}_resident_stamps.txt", "w") as fo:
            fo.write("# cycle stamps (s_memtime) per head of the logits-resident chunk step, -DEKR_PROFILE build: tools/experiments/exp_resident_prof.py\n")
            fo.write("".join(ln for ln in open(rs) if "amdgpu.ids" not in ln))
    for name, d in (("decode", dec["pmc"]), ("prefill", {k: {kk: vv["hbm_bytes_per_launch"] for kk, vv in v["kernels"].items()} for k, v in pre.items()})):
        print(name, json.dumps(d, indent=1)[:1800])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
