#!/usr/bin/env python3
"""Times the *reference* (its unmodified executor.py, predictor index 10 = TraceWeaverV3) on a generated corpus of the
bench workload's shape and writes the record bench.py prints as cpu_baseline.reference.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Needs /root/reference, so it runs in the build container (the GPU box has no
reference); the record (profiles/cpu_reference.json) is committed.  Stand-ins as in gen_golden.py: HiGHS for the absent
Gurobi (oracle/refrun/shims), two unused imports, the numpy-2-incompatible debug print.  The reference is
single-threaded: one core.

    python oracle/refrun/time_reference.py [--shape media|nodeio|alibaba] [--traces 1000] [--concurrency 1.6]

Shapes: `media` = a generated media-shape corpus (the headline bench workload's shape, --fix 2); `nodeio` = the shipped
corpus data/nodejs_microservices_with_arbitrary_file_io/node_1 as it is (BASELINE config 3, --fix 0, 1000 traces);
`alibaba` = a generated Alibaba-shape corpus through the --fix 5 route (BASELINE configs 4 / 5; the traces themselves are
not in the reference repository).  The record of every shape is kept in profiles/cpu_reference.json (`shapes`); the media
record also stays at the top level, where bench.py has always read it.

spans/s = (incoming + outgoing spans handed to FindAssignments) / (sum of the executor's own `--- s seconds ---` wall
per service, executor.py:1154,1189) -- the definition of SURVEY.md section 6 / 8(d).
"""
import argparse
import io
import json
import os
import pickle
import platform
import re
import runpy
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="media", choices=["media", "nodeio", "alibaba"])
    ap.add_argument("--traces", type=int, default=1000)
    ap.add_argument("--concurrency", type=float, default=None)
    args = ap.parse_args()
    from traceweaver_amd import synth

    conc = args.concurrency if args.concurrency is not None else {"media": 1.6, "nodeio": 0.0, "alibaba": 1.3}[args.shape]
    fix = {"media": "2", "nodeio": "0", "alibaba": "5"}[args.shape]
    if args.shape == "nodeio":   # the shipped corpus itself (symlinked files; the executor writes its file-order cache next to them)
        rel = "data/nodejs_microservices_with_arbitrary_file_io/node_1/"
        root = G.make_scratch_root(rel, max_files=args.traces)
    else:
        root = tempfile.mkdtemp(prefix="twreftime_")
        rel = "data/synthetic/%s_shape/" % args.shape
    try:
        if args.shape != "nodeio":
            os.symlink(os.path.join(G.REF, "src"), os.path.join(root, "src"))
            if args.shape == "media":
                synth.write_jaeger_corpus(os.path.join(root, rel), 11, args.traces, app=synth.MEDIA_APP, concurrency=conc,
                                          mean_service_us=4000.0, gap_us=300.0)
            else:
                synth.write_alibaba_corpus(os.path.join(root, rel), 5, args.traces, concurrency=conc)
            os.makedirs(os.path.join(root, "data", "misc"))
            with open(os.path.join(root, "data", "misc", "service_to_replica_new.pickle"), "wb") as fh:
                pickle.dump({}, fh)
            os.makedirs(os.path.join(root, "results"))
        pydir = os.path.join(root, "src", "trace_reconstructor", "ports", "python")
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = G.load_patched_v3(pydir)
        calls = []
        orig = v3mod.TraceWeaverV3.FindAssignments

        def timed(self, method, process, in_parts, out_parts, *a, **k):
            n = sum(len(v) for v in in_parts.values()) + sum(len(v) for v in out_parts.values())
            t0 = time.perf_counter()
            r = orig(self, method, process, in_parts, out_parts, *a, **k)
            calls.append((process, n, time.perf_counter() - t0, len(out_parts)))
            return r

        v3mod.TraceWeaverV3.FindAssignments = timed
        sys.argv = ["executor.py", "--relative_path", rel, "--compressed", "0", "--cache_rate", "0", "--fix", fix, "--test_name", "reftime",
                    "--load_level", "100", "--compress_factor", "1", "--repeat_factor", "1", "--execute_parallel", "0",
                    "--results_directory", os.path.join(root, "results") + "/", "--clear_cache", "1", "--predictor_indices", "10"]
        np.random.seed(10)
        buf, saved = io.StringIO(), sys.stdout
        sys.stdout = buf
        t0 = time.perf_counter()
        try:
            runpy.run_path(os.path.join(pydir, "executor.py"), run_name="__main__")
        finally:
            sys.stdout = saved
        whole = time.perf_counter() - t0
        acc = [float(x) for x in re.findall(r"End-to-end accuracy for method MaxScoreBatchSubsetWithSkips: ([0-9.]+)", buf.getvalue())]
        spans = sum(c[1] for c in calls)
        solve = sum(c[2] for c in calls)
        model = ""
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        rec = {"value": spans / solve, "unit": "spans/s", "cores": 1, "kind": "reference",
               "what": "the reference's executor.py (predictor index 10, TraceWeaverV3, both passes, HiGHS in place of Gurobi) on %s: "
                       "%d traces, services E = %s%s; spans handed to FindAssignments / sum of its wall times"
                       % ({"media": "a generated media-shape corpus", "nodeio": "the shipped nodejs_microservices_with_arbitrary_file_io/node_1 corpus (--fix 0)",
                           "alibaba": "a generated Alibaba-shape corpus (--fix 5 route)"}[args.shape],
                          args.traces, sorted((c[3] for c in calls), reverse=True), "" if args.shape == "nodeio" else ", concurrency %.1f" % conc),
               "shape": args.shape,
               "spans": spans, "find_assignments_s": solve, "whole_process_s": whole, "end_to_end_accuracy_pct": acc[-1] if acc else None,
               "per_service": [{"service": c[0], "spans": c[1], "seconds": c[2], "E": c[3]} for c in calls],
               "host": {"cpu": model, "cores_available": os.cpu_count(), "python": platform.python_version()},
               "measured_in": "build container (the GPU box has no /root/reference); single-threaded"}
        os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
        path = os.path.join(REPO, "profiles", "cpu_reference.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        shapes = dict(old.get("shapes", {}))
        if "shapes" not in old and old.get("value") is not None:
            shapes["media"] = {k: v for k, v in old.items() if k != "shapes"}
        shapes[args.shape] = rec
        top = dict(shapes.get("media", rec))
        top["shapes"] = shapes
        with open(path, "w") as fh:
            json.dump(top, fh, indent=1)
        print(json.dumps(rec, indent=1))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
