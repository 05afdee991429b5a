#!/usr/bin/env python3
"""Freeze golden vectors of the reference's `--fix 5` route (the output shape of alibaba-analysis/real-parser.py, which
exps/exp5 feeds to the executor), with and without `--compress_factor`.

TEST INFRASTRUCTURE ONLY; run by hand in the build container (needs /root/reference), outputs committed under
tests/golden/refali_*.npz.  The Alibaba traces themselves are not part of the reference repository, so the corpus is
generated (traceweaver_amd.synth.write_alibaba_corpus, seed and sizes recorded in the golden so that the test
regenerates the same files): rpc-id span ids, a server + client record per call, a service that calls itself and a few
traces that break parent-child containment.  The unmodified reference executor then parses it (ParseSpansJson with
first_span == None, executor.py:377-448) and runs predictor 10; recorded per service: the inputs the predictor was
handed (that is what the native ingest must reproduce), the call-order DAG, ground truth and the results.

The reference names the stand-in service of a self-call at random (helpers/misc.py:17-19); the golden keeps that name
in `process` / `out_eps`, the tests compare such names by their "-loop" suffix.
"""
import argparse
import os
import pickle
import runpy
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import gen_golden as G  # noqa: E402

REL = "data/alibaba_shape/call_graph_0/"
# (golden name, corpus seed, traces, concurrency, violations, compress factor, replicas per service)
CASES = [
    ("ali_s7", 7, 600, 1.6, 0.04, 1, {}),
    ("ali_s8_x3", 8, 500, 1.2, 0.03, 3, {"gw": 1, "cart": 2, "catalog": 1}),
]


def run(name, seed, n_traces, conc, viol, factor, replicas):
    from traceweaver_amd import synth

    root = tempfile.mkdtemp(prefix="twref_")
    os.symlink(os.path.join(G.REF, "src"), os.path.join(root, "src"))
    synth.write_alibaba_corpus(os.path.join(root, REL), seed, n_traces, concurrency=conc, violations=viol)
    os.makedirs(os.path.join(root, "data", "misc"))
    with open(os.path.join(root, "data", "misc", "service_to_replica_new.pickle"), "wb") as fh:
        pickle.dump({s: list(range(r)) for s, r in replicas.items()}, fh)
    os.makedirs(os.path.join(root, "results"))
    pydir = os.path.join(root, "src", "trace_reconstructor", "ports", "python")
    saved_path, saved_argv, saved_mods = list(sys.path), list(sys.argv), set(sys.modules)
    float_times = {}

    class Rec(G.Recorder):
        def install(self, v3mod):
            super().install(v3mod)
            V3 = v3mod.TraceWeaverV3
            inner = V3.FindAssignments

            def find(self_, method, process, in_parts, out_parts, *a, **k):
                in_ep = list(in_parts.keys())[0]
                out_eps = self_.GetOutEpsInOrder(out_parts, a[3])
                float_times[process] = (np.array([s.start_mus for s in in_parts[in_ep]], dtype=np.float64),
                                        [np.array([s.start_mus for s in out_parts[e]], dtype=np.float64) for e in out_eps],
                                        [s.GetId() for s in in_parts[in_ep]], self_)
                return inner(self_, method, process, in_parts, out_parts, *a, **k)

            V3.FindAssignments = find

    try:
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = G.load_patched_v3(pydir)
        rec = Rec()
        rec.install(v3mod)
        sys.argv = ["executor.py", "--relative_path", REL, "--compressed", "0", "--cache_rate", "0", "--fix", "5",
                    "--test_name", name, "--load_level", "100", "--compress_factor", str(factor), "--repeat_factor", "1",
                    "--execute_parallel", "0", "--results_directory", os.path.join(root, "results") + "/", "--clear_cache", "1",
                    "--predictor_indices", "10"]
        np.random.seed(G.SEED)
        saved_stdout, crashed = sys.stdout, None
        sys.stdout = open(os.devnull, "w")
        try:
            runpy.run_path(os.path.join(pydir, "executor.py"), run_name="__main__")
        except Exception as ex:   # the load-scaled route can stop in the refit (see gen_golden_compress.py)
            crashed = "%s: %s" % (type(ex).__name__, ex)
        finally:
            sys.stdout = saved_stdout
        if crashed:
            print(name, "reference run stopped in service", rec.cur["process"] if rec.cur else "?", "with", crashed, flush=True)
            c = rec.cur
            if c is not None and c["pass1_assign"] is not None and len(c["passes"]) == 1:
                E, n = len(c["out_eps"]), c["n_in"]
                c.update(final_parent=np.full((E, n), -9, np.int32), final_topk=np.full((E, n, G.TOPK), -9, np.int32),
                         not_best_count=-9, cnt_unassigned=-9, per_span_candidates=np.full(n, -9, np.int64), wall_s=np.nan,
                         windows=np.array(float_times[c["process"]][3].span_windows, dtype=np.int32).reshape(-1, 2), pass1_only=True)
                rec.services.append(c)
        for k, c in enumerate(rec.services):
            d = G.pack_service(name, c)
            fin, fout, ids, _ = float_times[c["process"]]
            d["pass1_only"] = np.array(1 if c.get("pass1_only") else 0)
            d["corpus"] = np.array([seed, n_traces], dtype=np.int64)
            d["corpus_params"] = np.array([conc, viol], dtype=np.float64)
            d["compress_factor"] = np.array(factor)
            d["replica_names"] = np.array(sorted(replicas))
            d["replica_counts"] = np.array([replicas[s] for s in sorted(replicas)], dtype=np.int64)
            d["service_order"] = np.array(k)
            d["in_trace_id"] = np.array([t for t, _ in ids])
            d["in_span_id"] = np.array([s for _, s in ids])
            if factor > 1:
                d["in_start"], d["out_start"] = fin, np.concatenate(fout)
            out = os.path.join(G.GOLDEN_DIR, "refali_%s__%d.npz" % (name, k))
            np.savez_compressed(out, **d)
            print("wrote", out, c["process"], "n_in", c["n_in"], "E", len(c["out_eps"]), flush=True)
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
    for case in CASES:
        if args.only and case[0] not in args.only:
            continue
        t0 = time.time()
        run(*case)
        print("%s done in %.0fs" % (case[0], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
