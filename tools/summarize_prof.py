#!/usr/bin/env python
"""Condense a rocprofv3 run (gpurun_out/<dir>/{trace,pmc_fetch,pmc_write}) into profiles/<tag>_*.

    python tools/summarize_prof.py gpurun_out/prof2 r01_decode

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB... on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled; WRITE_SIZE is taken as is.
"""
import csv
import json
import os
import sys


def main(src, tag):
    out = {}
    stats = os.path.join(src, "trace", [f for f in os.listdir(os.path.join(src, "trace")) if f.endswith("kernel_stats.csv")][0])
    rows = list(csv.DictReader(open(stats)))
    os.makedirs("profiles", exist_ok=True)
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if "ekv_" in r["Name"]:
                w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
                out.setdefault("kernels", {})[r["Name"]] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                                               min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3)
    for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        d = os.path.join(src, name)
        if not os.path.isdir(d):
            continue
        f = os.path.join(d, [x for x in os.listdir(d) if x.endswith("counter_collection.csv")][0])
        per = {}
        for r in csv.DictReader(open(f)):
            if "ekv_" in r["Kernel_Name"] and r["Counter_Name"] == key:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            out.setdefault("pmc", {}).setdefault(k, {})[key + "_KiB_avg"] = sum(v) / len(v)
            out["pmc"][k]["VGPR_SGPR_note"] = "see counter_collection.csv columns VGPR_Count / SGPR_Count"
    for k, v in out.get("pmc", {}).items():
        if "FETCH_SIZE_KiB_avg" in v and "WRITE_SIZE_KiB_avg" in v:
            v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024.0
            v["correction"] = "2 x FETCH_SIZE (gfx950 wide coalesced reads tallied at half) + WRITE_SIZE, KiB -> bytes"
    bj = os.path.join(src, "bench_trace.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        out["bench_line_under_rocprof"] = json.loads(open(bj).read().strip().splitlines()[-1])
    json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
    print(json.dumps(out.get("pmc", {}), indent=1)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
