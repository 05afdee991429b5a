"""Throughput of the default workload against the size of the resident batch (replicas of the media service graph per
GPU).  The enumeration / selection kernels end in tails as long as their longest single item (a 25-span window, a
4e4-tuple span), which do not grow with the batch: a larger resident batch amortises them.  No torch (a fresh box
spends a minute or two importing it): numpy + ctypes on the C-ABI only.

    python profiles/tools/batch_sweep.py --replicas 4,8,16,32 --out gpurun_out/batch_sweep.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", default="4,8,16,32")
    ap.add_argument("--n-in", type=int, default=100000)
    ap.add_argument("--concurrency", type=float, default=1.6)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--workload", default="media", choices=["media", "nodejs"])
    ap.add_argument("--tag", default=None, help="label copied into every row (e.g. the build variant)")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from traceweaver_amd import synth
    from traceweaver_amd.engine import Engine

    eng = Engine(0, lib_path=args.lib)
    rows = []
    for rep in [int(x) for x in args.replicas.split(",")]:
        if args.workload == "nodejs":
            units, truth = synth.make_nodejs_workload(1000, args.n_in, concurrency=args.concurrency, replicas=rep)
        else:
            units, truth = synth.make_workload(1000, args.n_in, services=synth.MEDIA_SERVICES, replicas=rep, concurrency=args.concurrency)
        spans = sum(u.n_spans for u in units)
        t0 = time.perf_counter()
        eng.load(units)
        eng.set_truth(truth)
        load_s = time.perf_counter() - t0

        def step():
            eng.run_pass1()
            t1 = eng.timing()
            eng.fit_mixtures()
            eng.run_pass2()
            t2 = eng.timing()
            return t1, t2, eng.evaluate()

        step()
        t0 = time.perf_counter()
        enum, sel, fit, rounds = [], [], [], []
        for _ in range(args.steps):
            t1, t2, res = step()
            enum += [t1["enumerate"], t2["enumerate"]]
            sel += [t1["select"], t2["select"]]
            fit += [t2["fit"]]
        dt = (time.perf_counter() - t0) / args.steps
        stats = eng.results(2, fields=("unit_stats",))
        row = {"tag": args.tag, "workload": args.workload, "replicas": rep, "spans": spans, "ms_per_step": dt * 1e3, "spans_per_s": spans / dt, "load_s": load_s,
               "enumerate_ms": float(np.mean(enum)), "select_ms": float(np.mean(sel)), "fit_ms": float(np.mean(fit)),
               "frac_hbm_8TBps": 20.0 * spans / (float(np.mean(enum)) * 1e-3) / 8e12,
               "accuracy": float(np.mean([r["accuracy"] for r in res])),
               "budget_windows": int(sum(r["budget_windows"] for r in stats))}
        rows.append(row)
        print(json.dumps(row), flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            with open(args.out, "a") as f:
                f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
