#!/usr/bin/env python3
"""Freeze golden vectors of the reference's *load-scaling* path (`--compress_factor N`, the route exps/exp5 takes).

TEST INFRASTRUCTURE ONLY; run by hand in the build container (needs /root/reference), outputs committed under
tests/golden/refcmp_*.npz.  Same machinery as gen_golden.py (the unmodified reference runs, wrapped to record).

What is different from gen_golden.py
------------------------------------
* `data/misc/service_to_replica_new.pickle` lists one replica for every service of the corpus, so the executor's
  per-service load factor is max(1, ceil(compress_factor / 1)) = compress_factor (executor.py:1086-1097).
* executor.py:1146-1148 then calls transforms.repeat_change_spans (helpers/transforms.py:10-40): partitions sorted by
  trace id, incoming start x = start / load_factor (a Python float), outgoing start = x + (out.start - in.start),
  durations untouched, partitions re-sorted by (start, start + duration).  Every timestamp the predictor sees is a
  float64 from there on; the recorded `in_start` / `out_start` are therefore float64 and the durations int64
  (end times are formed as fl(start + duration) wherever the reference needs them).
* Known hazards of the reference on this path, visible in the frozen runs: ComputeEpPairDistParams5
  (traceweaver_v3.py:717-762) looks the *outgoing* span up in `self.all_spans` -- the untransformed originals -- and
  subtracts the transformed incoming timestamp from it, so the mixtures pass 2 scores with are fitted to differences
  of the order of the epoch (pass-1 accuracy 99 %, final accuracy 0 % on hotel_load50 x2); on some corpora every
  GaussianMixture fit of such a row raises and the run stops at traceweaver_v3.py:780 (argmin of an empty list).
  Services that completed before such a stop are still written, and so is the pass-1 record of the service it
  stopped in (`pass1_only` = 1; the fields of the final result hold -9).
"""
import argparse
import json
import os
import pickle
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

# (golden name, relative data dir, --fix, compress factor)
DATASETS = [
    ("hotel_load50_x2", "data/hotel_reservation/hotel_load50/", 2, 2),
    ("hotel_load25_x3", "data/hotel_reservation/hotel_load25/", 2, 3),
    ("node_load25_x3", "data/nodejs_microservices/node_load25/", 0, 3),
    ("media_load25_x2", "data/media_microservices/media_load25/", 1, 2),
    ("hotel_load100_x4", "data/hotel_reservation/hotel_load100/", 2, 4),
]


def services_of(rel_dir):
    names = set()
    d = os.path.join(G.REF, rel_dir)
    for f in sorted(os.listdir(d))[:50]:
        if f.endswith(".json"):
            with open(os.path.join(d, f)) as fh:
                for t in json.load(fh)["data"]:
                    for p in t.get("processes", {}).values():
                        names.add(p["serviceName"])
    return names


def run(name, rel_dir, fix, factor):
    float_times = {}

    class Rec(G.Recorder):
        def install(self, v3mod):
            super().install(v3mod)
            V3 = v3mod.TraceWeaverV3
            inner = V3.FindAssignments

            def find(self_, method, process, in_parts, out_parts, *a, **k):
                graph = a[3]
                in_ep = list(in_parts.keys())[0]
                out_eps = self_.GetOutEpsInOrder(out_parts, graph)
                float_times[process] = (
                    np.array([s.start_mus for s in in_parts[in_ep]], dtype=np.float64),
                    [np.array([s.start_mus for s in out_parts[e]], dtype=np.float64) for e in out_eps],
                    all(isinstance(s.start_mus, float) for s in in_parts[in_ep]),
                    [s.trace_id for s in in_parts[in_ep]],
                    self_,
                )
                return inner(self_, method, process, in_parts, out_parts, *a, **k)

            V3.FindAssignments = find

    root = G.make_scratch_root(rel_dir)
    with open(os.path.join(root, "data", "misc", "service_to_replica_new.pickle"), "wb") as fh:
        pickle.dump({s: [0] for s in services_of(rel_dir)}, fh)
    pydir = os.path.join(root, "src", "trace_reconstructor", "ports", "python")
    saved_path, saved_argv, saved_mods = list(sys.path), list(sys.argv), set(sys.modules)
    import runpy
    import shutil
    try:
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = G.load_patched_v3(pydir)
        rec = Rec()
        rec.install(v3mod)
        sys.argv = [
            "executor.py", "--relative_path", rel_dir, "--compressed", "0", "--cache_rate", "0",
            "--fix", str(fix), "--test_name", name, "--load_level", "100", "--compress_factor", str(factor),
            "--repeat_factor", "1", "--execute_parallel", "0",
            "--results_directory", os.path.join(root, "results") + "/", "--clear_cache", "1",
            "--predictor_indices", "10",
        ]
        np.random.seed(G.SEED)
        saved_stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")
        crashed = None
        try:
            runpy.run_path(os.path.join(pydir, "executor.py"), run_name="__main__")
        except Exception as ex:  # the reference itself can fail on this path (see the module docstring); keep what completed
            crashed = "%s: %s" % (type(ex).__name__, ex)
        finally:
            sys.stdout = saved_stdout
        res = os.path.join(root, "results")
        acc = {"MaxScoreBatchSubsetWithSkips": np.nan, "MaxScoreBatchSubsetWithSkipsTopK": np.nan}
        if crashed is None:
            with open(os.path.join(res, [f for f in os.listdir(res) if f.startswith("accuracy_")][0]), "rb") as fh:
                acc = pickle.load(fh)
        else:
            print(name, "reference run stopped in service", rec.cur["process"] if rec.cur else "?", "with", crashed, flush=True)
            c = rec.cur
            if c is not None and c["pass1_assign"] is not None and len(c["passes"]) == 1:
                # the stop came in the refit after pass 1: everything pass 1 produced is on record -- keep it, marked
                E, n = len(c["out_eps"]), c["n_in"]
                c.update(final_parent=np.full((E, n), -9, np.int32), final_topk=np.full((E, n, G.TOPK), -9, np.int32),
                         not_best_count=-9, cnt_unassigned=-9, per_span_candidates=np.full(n, -9, np.int64), wall_s=np.nan,
                         windows=np.array(float_times[c["process"]][4].span_windows, dtype=np.int32).reshape(-1, 2), pass1_only=True)
                rec.services.append(c)
        for c in rec.services:
            d = G.pack_service(name, c)
            d["pass1_only"] = np.array(1 if c.get("pass1_only") else 0)
            fin, fout, is_float, tids = float_times[c["process"]][:4]
            assert is_float
            d["in_start"] = fin                       # float64: what the predictor saw
            d["out_start"] = np.concatenate(fout)
            d["compress_factor"] = np.array(factor)
            d["in_trace_id"] = np.array(tids)
            d["e2e_accuracy"] = np.array(acc["MaxScoreBatchSubsetWithSkips"])
            d["e2e_topk_accuracy"] = np.array(acc["MaxScoreBatchSubsetWithSkipsTopK"])
            out = os.path.join(G.GOLDEN_DIR, "refcmp_%s__%s.npz" % (name, c["process"]))
            np.savez_compressed(out, **d)
            print("wrote", out, "n_in", c["n_in"], "E", len(c["out_eps"]), "wall %.1fs" % c["wall_s"], flush=True)
        print(name, "e2e accuracy", acc, flush=True)
    finally:
        sys.path[:] = saved_path
        sys.argv[:] = saved_argv
        for m in set(sys.modules) - saved_mods:
            del sys.modules[m]
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    for name, rel, fix, factor in DATASETS:
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        run(name, rel, fix, factor)
        print("%s done in %.0fs" % (name, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
