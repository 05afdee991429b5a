"""profiles/<tag>_regimes.md + profiles/<tag>_regimes/*_kernel_stats_top.csv from the output of `bash profiles/tools/regimes.sh <tag> stats`
(gpurun_out/<tag>_<regime>/p_kernel_stats.csv and the bench line at the end of gpurun_out/<tag>_<regime>.log).

    python profiles/tools/regimes_md.py r03
"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REGIMES = [("nodejs", "config 3 shape (`--workload nodejs --n-in 20000`)"),
           ("alibaba", "config 4: Alibaba 1 M-span slice on one GPU (`--workload alibaba`)"),
           ("c4", "media shape, concurrency 4 (`--concurrency 4 --n-in 20000 --replicas 4`)"),
           ("c8", "media shape, concurrency 8 (`--concurrency 8 --n-in 5000 --replicas 4`)")]


def short(name):
    n = name.replace("void ", "").replace("tw::", "")
    for cut in ("(tw::Dev", "(Dev", "(tw::FitDev", "(FitDev"):
        if cut in n:
            n = n[:n.index(cut)]
    return n[:70]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(REPO, "gpurun_out")
    out_dir = os.path.join(REPO, "profiles", tag + "_regimes")
    os.makedirs(out_dir, exist_ok=True)
    lines = ["# %s load regimes: which kernels a step is made of (one MI355X, kernels of HEAD)\n" % tag,
             "`bash profiles/tools/regimes.sh %s stats`: `rocprofv3 --kernel-trace --stats -- python bench.py <args> --cpu-sample 0 --regimes 0 --sync engine "
             "--steps 2 --warmup 1` per regime (3 steps = 6 passes in the trace); the five longest kernels each, and the bench line of the profiled "
             "run (rocprofv3 attached: a few per cent slower than the `regimes` block of the default `python bench.py` line).\n" % tag]
    for key, title in REGIMES:
        stats = os.path.join(src, "%s_%s" % (tag, key), "p_kernel_stats.csv")
        log = os.path.join(src, "%s_%s.log" % (tag, key))
        if not os.path.exists(stats):
            continue
        rows = [r for r in csv.DictReader(open(stats)) if "k_copy16" not in r["Name"]]   # (tw_measure_hbm_copy: the ceiling quoted in the bench line, not part of a step)
        with open(os.path.join(out_dir, key + "_kernel_stats_top.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows[:25])
        bench = None
        if os.path.exists(log):
            js = [l for l in open(log).read().split("\n") if l.startswith("{")]
            bench = json.loads(js[-1]) if js else None
        lines.append("## %s\n" % title)
        if bench:
            g = bench["roofline"]["group_ms_per_launch"]
            lines.append("%.3g spans/s, %.1f ms per step, accuracy %.4f, windows not proven optimal %d; per launch: enumerate %.2f, select %.2f, repair %.2f ms, refit %.2f ms per step\n"
                         % (bench["value"], bench["ms_per_step"], bench["accuracy"], bench["budget_windows"], g["k_enumerate"], g["k_select"], g["k_repair"], g["k_fit"]))
        lines.append("| kernel | calls | avg ms | max ms | % of GPU time |\n|---|---|---|---|---|")
        for r in rows[:5]:
            lines.append("| `%s` | %s | %.2f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
        lines.append("")
    open(os.path.join(REPO, "profiles", tag + "_regimes.md"), "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
