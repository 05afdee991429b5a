#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/collect.sh into the committed per-round summary.

    python profiles/summarize.py r01

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats table), profiles/<tag>_traffic.json
(HBM bytes per launch per kernel group, read by bench.py for roofline.traffic) and profiles/<tag>_summary.md.

Units / corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads, so reads are doubled ("x2
corrected"); our access pattern is mostly narrow / scattered, for which the counter is uncalibrated, so both
the raw and the corrected figure are kept.  WRITE_SIZE is uncalibrated and taken as is.
"""
import collections
import csv
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def group_of(name):
    if "k_enumerate" in name:
        return "k_enumerate"
    if "k_select" in name:
        return "k_select"
    if "k_fit" in name:
        return "k_fit"
    if "rocprim" in name or "k_rows_" in name or "k_key_bits" in name:
        return "sort"
    if "k_block_params" in name:
        return "params"
    if "k_evaluate" in name or "k_count_flags" in name:
        return "evaluate"
    if any(k in name for k in ("k_scan", "k_perfect_cut", "k_window")):
        return "windows"
    if any(k in name for k in ("k_claim", "k_detect", "k_repair")):
        return "repair"
    return "other"


def counter_per_group(path, counter):
    per, calls = collections.defaultdict(float), collections.defaultdict(int)
    first_kernel_calls = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        g = group_of(r["Kernel_Name"])
        per[g] += float(r["Counter_Value"]) * 1024.0
        calls[r["Kernel_Name"]] += 1
    return per, calls


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out = os.path.join(REPO, "profiles")
    src = os.path.join(REPO, "gpurun_out")
    stats = os.path.join(src, tag + "_stats", tag + "_kernel_stats.csv")
    shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
    bench = json.loads([l for l in open(os.path.join(src, tag + "_stats.log")).read().split("\n") if l.startswith("{")][-1])
    passes = 2 * (bench["steps"] + bench["warmup"])           # enumerate / select launches: one set per pass
    fetch, _ = counter_per_group(os.path.join(src, tag + "_fetch", tag + "_counter_collection.csv"), "FETCH_SIZE")
    write, _ = counter_per_group(os.path.join(src, tag + "_write", tag + "_counter_collection.csv"), "WRITE_SIZE")
    rows = list(csv.DictReader(open(stats)))
    dur = collections.defaultdict(float)
    for r in rows:
        dur[group_of(r["Name"])] += float(r["TotalDurationNs"])
    spans = bench["config"]["spans_per_gpu"]
    traffic = {"tag": tag, "workload": bench["config"]["workload"], "spans_per_launch": spans, "launch_sets": passes, "groups": {}}
    for g in sorted(dur, key=lambda k: -dur[k]):
        n = passes if g in ("k_enumerate", "k_select", "repair") else passes // 2   # per pass / per step
        traffic["groups"][g] = {
            "avg_ms_per_launch_set": dur[g] / n / 1e6,
            "fetch_bytes_raw": fetch.get(g, 0.0) / n, "fetch_bytes_x2": 2 * fetch.get(g, 0.0) / n,
            "write_bytes": write.get(g, 0.0) / n,
        }
    json.dump(traffic, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)
    with open(os.path.join(out, tag + "_summary.md"), "w") as f:
        f.write("# %s rocprofv3 summary\n\n`%s`\n\nbench line: %.3g %s, %.1f ms/step, accuracy %.4f\n\n" % (
            tag, "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --cpu-sample 0", bench["value"], bench["unit"],
            bench["ms_per_step"], bench["accuracy"]))
        f.write("| kernel group | avg ms per launch set | algorithmic MB | HBM read MB (raw / x2) | HBM write MB |\n|---|---|---|---|---|\n")
        for g, v in traffic["groups"].items():
            f.write("| %s | %.3f | %.1f | %.1f / %.1f | %.1f |\n" % (g, v["avg_ms_per_launch_set"], 20.0 * spans / 1e6 if g == "k_enumerate" else float("nan"),
                                                               v["fetch_bytes_raw"] / 1e6, v["fetch_bytes_x2"] / 1e6, v["write_bytes"] / 1e6))
        f.write("\nTop kernels (rocprofv3 --stats):\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
        for r in rows[:14]:
            f.write("| `%s` | %s | %.1f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print(open(os.path.join(out, tag + "_summary.md")).read())


if __name__ == "__main__":
    main()
