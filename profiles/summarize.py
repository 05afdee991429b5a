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
    if "k_copy16" in name:
        return "hbm_copy_measurement"   # tw_measure_hbm_copy: the measured ceiling quoted in the bench line, not part of a step
    if "k_enumerate" in name or "k_classify" in name or "k_merge_parts" in name:
        return "k_enumerate"   # the group bench.py times between its events: cut-offs / lists, both enumeration kernels, merge of split spans
    if "k_select" in name:
        return "k_select"
    if "k_fit" in name:
        return "k_fit"
    if "rocprim" in name or "k_rows_" in name or "k_key_bits" in name or "k_rank_ends" in name or "k_place_ends" in name:
        return "sort"
    if "k_block_params" in name:
        return "params"
    if "k_evaluate" in name or "k_count_flags" in name:
        return "evaluate"
    if any(k in name for k in ("k_scan", "k_perfect_cut", "k_window", "k_cut_scan", "k_flags_scan", "k_index_fix")):
        return "windows"
    if any(k in name for k in ("k_claim", "k_detect", "k_repair")):
        return "repair"
    return "other"


def counter_per_group(path, counter):
    per, calls = collections.defaultdict(float), collections.defaultdict(int)
    if not os.path.exists(path):   # (a quick look without the FETCH / WRITE passes: the traffic columns stay zero)
        return per, calls
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        g = group_of(r["Kernel_Name"])
        per[g] += float(r["Counter_Value"]) * 1024.0
        calls[r["Kernel_Name"]] += 1
    return per, calls


SIMDS = 256 * 4  # MI355X: 256 CUs x 4 SIMDs
XCDS = 8         # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (measured: 19.1 counts per ns of kernel time = 8 x 2.39 GHz)


def short(name):
    n = name.replace("void ", "").replace("tw::", "")
    for cut in ("(tw::Dev", "(Dev", "(tw::FitDev", "(FitDev"):
        if cut in n:
            n = n[:n.index(cut)]
    return n[:60]


def counters_per_kernel(path):
    """{kernel: {counter: sum over dispatches}} plus per-kernel launch resources from one counter_collection.csv"""
    per, res = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    if not os.path.exists(path):
        return per, res
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            per[k]["_dispatches"] += 1
            per[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        res[k] = (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]),
                  int(r["Workgroup_Size"]))
    return per, res


def counter_report(tag, src, out, rows):
    """Per-kernel SQ / GRBM counters of the sq1, sq2, grbm passes -> profiles/<tag>_counters.md.  Formulas are the
    gfx94x derived metrics (ROCm 7.2 ships none for gfx950, MI355X_MICROARCH.md): SQ cycle counters tick in quad-cycles.
      waves/SIMD   = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs)        mean resident wavefronts per SIMD
      VALU busy    = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE * 1024)         share of SIMD cycles issuing VALU work
      lanes active = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)         (divergence: 1.0 = all 64 lanes)
      wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
      LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    """
    sq1, res = counters_per_kernel(os.path.join(src, tag + "_sq1", tag + "_counter_collection.csv"))
    sq2, _ = counters_per_kernel(os.path.join(src, tag + "_sq2", tag + "_counter_collection.csv"))
    grbm, _ = counters_per_kernel(os.path.join(src, tag + "_grbm", tag + "_counter_collection.csv"))
    if not sq1:
        return
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(out, tag + "_counters.md"), "w") as f:
        f.write("# %s per-kernel counters (rocprofv3 --pmc, one pass per counter set, same bench command)\n\n" % tag)
        f.write(counter_report.__doc__.split("Formulas", 1)[1].join(["Formulas", ""]) if False else "")
        f.write("Formulas (gfx94x derived metrics; SQ cycle counters tick in quad-cycles): waves/SIMD = 4*SQ_WAVE_CYCLES/(cycles*1024) with cycles = GRBM_GUI_ACTIVE/8 (the counter is summed over the 8 XCDs); "
                "VALU busy = 4*SQ_ACTIVE_INST_VALU/(cycles*1024); lanes = SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU); "
                "wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; LDS conflict = "
                "SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE; VGPR = arch + accumulation registers per lane; scratch in bytes per lane.\n\n")
        f.write("| kernel | % GPU time | avg us | VGPR | LDS B | scratch | waves launched/call | waves/SIMD | VALU busy | lanes | wait | stall | issue | "
                "LDS conflict | VALU : SALU : LDS : VMEM rd : VMEM wr insts/wave |\n|" + "---|" * 15 + "\n")
        for r in rows[:24]:
            k = short(r["Name"])
            a, b, g = sq1.get(k), sq2.get(k), grbm.get(k)
            if not a or a["_dispatches"] == 0:
                continue
            n = a["_dispatches"]
            gui = (g["GRBM_GUI_ACTIVE"] / g["_dispatches"] / XCDS) if g and g["_dispatches"] else float("nan")
            wc = a["SQ_WAVE_CYCLES"] / n
            waves = a["SQ_WAVES"] / n
            v = res.get(k, (0, 0, 0, 0, 0))
            nb = b["_dispatches"] if b and b["_dispatches"] else float("nan")
            lanes = (a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"])) if a.get("SQ_THREAD_CYCLES_VALU") and a["SQ_ACTIVE_INST_VALU"] else float("nan")
            mix = "-"
            if b:
                wv = max(a["SQ_WAVES"] / n, 1.0)
                mix = "%.0f : %.0f : %.0f : %.0f : %.0f" % (a["SQ_INSTS_VALU"] / n / wv, b["SQ_INSTS_SALU"] / nb / wv, b["SQ_INSTS_LDS"] / nb / wv,
                                                       b["SQ_INSTS_VMEM_RD"] / nb / wv, b["SQ_INSTS_VMEM_WR"] / nb / wv)
            conflict = (b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"]) if b and b["SQ_LDS_IDX_ACTIVE"] else float("nan")
            f.write("| `%s` | %.1f | %.1f | %d | %d | %d | %.0f | %.2f | %.3f | %.2f | %.2f | %.2f | %.2f | %.3f | %s |\n" % (
                k, 100.0 * float(r["TotalDurationNs"]) / total, float(r["AverageNs"]) / 1e3, v[0], v[2], v[3], waves,
                4.0 * wc / (gui * SIMDS), 4.0 * a["SQ_ACTIVE_INST_VALU"] / n / (gui * SIMDS), lanes,
                a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1), a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1),
                a["SQ_ACTIVE_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1), conflict, mix))
    print(open(os.path.join(out, tag + "_counters.md")).read())


def _pass_end(name):
    """Kernels that follow a pass: the refit's first kernel (after pass 1) and the accuracy reduction (after pass 2)."""
    return "k_fit_runs" in name or "k_fit_compress" in name or "k_evaluate" in name


def enumerate_wall(trace_path):
    """Wall-clock span of the enumeration group per pass (first start to last end of its kernels in the pass), from the kernel trace.
    The group's kernels run on one stream per endpoint-count class and overlap, so the *sum* of their durations (the `avg ms per launch
    set` column) exceeds this span; since round 6 the window / selection stages of the classes that are done run inside the span too
    (DESIGN.md 3.2).  bench.py's HIP events around the group measure the same span.  Passes are told apart by what follows them
    (k_fit_runs after pass 1, k_evaluate after pass 2).  Returns (mean span ms, mean of pass 1, mean of pass 2) over the spans
    longer than 1 ms, or None."""
    if not os.path.exists(trace_path):
        return None
    rows = sorted(csv.DictReader(open(trace_path)), key=lambda r: int(r["Start_Timestamp"]))
    sets, cur = [], None
    for r in rows:
        g = group_of(r["Kernel_Name"])
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if g == "k_enumerate":
            cur = [s, e] if cur is None else [min(cur[0], s), max(cur[1], e)]
        elif _pass_end(r["Kernel_Name"]) and cur is not None:
            sets.append(cur)
            cur = None
    if cur:
        sets.append(cur)
    spans = [(c[1] - c[0]) / 1e6 for c in sets if c[1] - c[0] > 1e6]
    if not spans:
        return None
    return sum(spans) / len(spans), sum(spans[0::2]) / max(len(spans[0::2]), 1), sum(spans[1::2]) / max(len(spans[1::2]), 1)


def timeline(trace_path):
    """Kernels of the last pass in the trace (pass 2 of the last step: from the first enumeration kernel after the refit to the
    accuracy reduction), enumeration, windows, selection and consumption, in start order: (short name, queue, start offset us,
    duration us) -- which kernels overlap and which chain ends the pass."""
    if not os.path.exists(trace_path):
        return None
    rows = sorted(csv.DictReader(open(trace_path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "k_evaluate" in r["Kernel_Name"]]
    if not ends:
        return None
    last = ends[-1]
    first = last
    while first > 0 and "k_fit" not in rows[first - 1]["Kernel_Name"] and "k_mix_consts" not in rows[first - 1]["Kernel_Name"]:
        first -= 1
    out, t0 = [], None
    for r in rows[first:last]:
        g = group_of(r["Kernel_Name"])
        if g not in ("k_enumerate", "k_select", "windows", "repair") and "k_finalize" not in r["Kernel_Name"]:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        t0 = s if t0 is None else t0
        out.append((short(r["Kernel_Name"]), r.get("Queue_Id", ""), (s - t0) / 1e3, (e - s) / 1e3))
    return out


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
    digest = os.path.join(src, tag + "_digest.txt")
    traffic = {"tag": tag, "source_digest": open(digest).read().strip() if os.path.exists(digest) else None,
               "commit": open(os.path.join(src, tag + "_commit.txt")).read().strip() if os.path.exists(os.path.join(src, tag + "_commit.txt")) else None,
               "workload": bench["config"]["workload"], "spans_per_launch": spans, "launch_sets": passes, "groups": {}}
    for g in sorted(dur, key=lambda k: -dur[k]):
        n = passes if g in ("k_enumerate", "k_select", "repair") else passes // 2   # per pass / per step
        traffic["groups"][g] = {
            "avg_ms_per_launch_set": dur[g] / n / 1e6,
            "fetch_bytes_raw": fetch.get(g, 0.0) / n, "fetch_bytes_x2": 2 * fetch.get(g, 0.0) / n,
            "write_bytes": write.get(g, 0.0) / n,
        }
    wall = enumerate_wall(os.path.join(src, tag + "_stats", tag + "_kernel_trace.csv"))
    if wall:
        traffic["groups"]["k_enumerate"]["wall_ms_per_launch_set"] = wall[0]
        traffic["groups"]["k_enumerate"]["wall_ms_pass1"], traffic["groups"]["k_enumerate"]["wall_ms_pass2"] = wall[1], wall[2]
    json.dump(traffic, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)
    with open(os.path.join(out, tag + "_summary.md"), "w") as f:
        f.write("# %s rocprofv3 summary\n\n`%s`\n\nbench line: %.3g %s, %.1f ms/step, accuracy %.4f\n\n" % (
            tag, "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --cpu-sample 0", bench["value"], bench["unit"],
            bench["ms_per_step"], bench["accuracy"]))
        f.write("| kernel group | avg ms per launch set | algorithmic MB | HBM read MB (raw / x2) | HBM write MB |\n|---|---|---|---|---|\n")
        for g, v in traffic["groups"].items():
            f.write("| %s | %.3f | %.1f | %.1f / %.1f | %.1f |\n" % (g, v["avg_ms_per_launch_set"], 20.0 * spans / 1e6 if g == "k_enumerate" else float("nan"),
                                                               v["fetch_bytes_raw"] / 1e6, v["fetch_bytes_x2"] / 1e6, v["write_bytes"] / 1e6))
        if wall:
            f.write("\n`k_enumerate`: its kernels run on one stream per endpoint-count class and overlap -- the column above is the *sum* of their "
                    "durations; the wall-clock span of a launch set in the kernel trace is **%.2f ms** (pass 1: %.2f, pass 2: %.2f), the HIP events of "
                    "`bench.py` around the same launches read %.2f ms (`roofline.kernel_ms`).\n" % (
                        wall[0], wall[1], wall[2], bench.get("roofline", {}).get("kernel_ms", float("nan"))))
        tl = timeline(os.path.join(src, tag + "_stats", tag + "_kernel_trace.csv"))
        if tl:
            f.write("\nTimeline of the last pass (pass 2 of the last step): every class on its own queue, enumeration then its window / selection / consumption stage (DESIGN.md 3.2); kernels on different queues overlap:\n\n"
                    "| kernel | queue | start us | duration us |\n|---|---|---|---|\n")
            for name, q, st, du in tl:
                if du >= 20.0:
                    f.write("| `%s` | %s | %.0f | %.0f |\n" % (name[:60], q, st, du))
        f.write("\nTop kernels (rocprofv3 --stats):\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
        for r in rows[:14]:
            f.write("| `%s` | %s | %.1f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print(open(os.path.join(out, tag + "_summary.md")).read())
    counter_report(tag, src, out, rows)


if __name__ == "__main__":
    main()
