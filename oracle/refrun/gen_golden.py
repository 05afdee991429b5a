#!/usr/bin/env python3
"""Freeze golden vectors by running the *reference* TraceWeaver code in this container.

TEST INFRASTRUCTURE ONLY.  Needs /root/reference (read-only) and is therefore run by hand in the
build container, never on the GPU box; its outputs (tests/golden/*.npz) are committed.

What it does
------------
1. Builds a scratch project root under $TMPDIR that *symlinks* the reference sources
   (`src/` -> /root/reference/src) and one data directory (a real directory of symlinked JSON
   files, because executor.py:321-337 writes `time_order_filenames.pickle` next to the data), plus
   an empty `data/misc/service_to_replica_new.pickle` (loaded unconditionally, executor.py:912).
   Nothing from the reference is copied into this repository.
2. Puts three stand-ins on sys.path (oracle/refrun/shims): `gurobi_optimods.mwis` (exact MWIS via
   HiGHS; Gurobi is unobtainable), `pygmmis`, `deepdiff` (imported but unused by the reference).
3. Loads `algorithms/traceweaver_v3.py` through a loader that blanks the print-only debug block at
   traceweaver_v3.py:788-794 (it compares an ndarray with `[]` and raises under numpy 2), and casts
   the input of GaussianMixture.score to float64 (scikit-learn 1.7 truncates integer inputs; the
   reference pins 1.5.1, which does not).
4. Wraps (does not edit) TraceWeaverV3.{FindAssignments, FindTopKAssignments, GetAssignmentsMIS,
   ComputeEpPairDistParams3, ComputeEpPairDistParams5} to record inputs, intermediates and
   outputs as index arrays, seeds numpy's global RNG (the reference's GMM model selection draws
   from it, traceweaver_v3.py:774) and runs executor.py (predictor index 10) with runpy.
5. Writes one compressed .npz per (dataset, service).

Usage:  python oracle/refrun/gen_golden.py [--only NAME ...]
"""
import argparse
import importlib.util
import os
import pickle
import runpy
import shutil
import sys
import tempfile
import time

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
SEED = 10
TOPK = 5

# (name, relative data dir, --fix, max files)
DATASETS = [
    ("hotel_load100", "data/hotel_reservation/hotel_load100/", 2),
    ("hotel_load150", "data/hotel_reservation/hotel_load150/", 2),
    ("media_load100", "data/media_microservices/media_load100/", 1),
    ("media_load150", "data/media_microservices/media_load150/", 1),
    ("nodeio_1", "data/nodejs_microservices_with_arbitrary_file_io/node_1/", 0),
    ("nodeio_0.6", "data/nodejs_microservices_with_arbitrary_file_io/node_0.6/", 0),
    ("node_load150", "data/nodejs_microservices/node_load150/", 0),
    # lighter load levels of the same applications
    ("hotel_load50", "data/hotel_reservation/hotel_load50/", 2),
    ("media_load50", "data/media_microservices/media_load50/", 1),
    ("node_load100", "data/nodejs_microservices/node_load100/", 0),
    ("nodeio_0.2", "data/nodejs_microservices_with_arbitrary_file_io/node_0.2/", 0),
    # the remaining load levels shipped with the reference (media_load75 holds 1500 files: the first 1000 by name)
    ("hotel_load25", "data/hotel_reservation/hotel_load25/", 2), ("hotel_load75", "data/hotel_reservation/hotel_load75/", 2),
    ("hotel_load125", "data/hotel_reservation/hotel_load125/", 2),
    ("media_load25", "data/media_microservices/media_load25/", 1), ("media_load125", "data/media_microservices/media_load125/", 1),
    ("media_load75", "data/media_microservices/media_load75/", 1),   # 1500 files: make_scratch_root keeps the first 1000 by name (reference hazard H3)
    ("node_load25", "data/nodejs_microservices/node_load25/", 0), ("node_load50", "data/nodejs_microservices/node_load50/", 0),
    ("node_load75", "data/nodejs_microservices/node_load75/", 0), ("node_load125", "data/nodejs_microservices/node_load125/", 0),
    ("nodeio_0", "data/nodejs_microservices_with_arbitrary_file_io/node_0/", 0),
    ("nodeio_0.4", "data/nodejs_microservices_with_arbitrary_file_io/node_0.4/", 0),
    ("nodeio_0.8", "data/nodejs_microservices_with_arbitrary_file_io/node_0.8/", 0),
]


def make_scratch_root(rel_dir, max_files=1000):
    root = tempfile.mkdtemp(prefix="twref_")
    os.symlink(os.path.join(REF, "src"), os.path.join(root, "src"))
    src_dir = os.path.join(REF, rel_dir)
    dst_dir = os.path.join(root, rel_dir)
    os.makedirs(dst_dir)
    files = sorted(f for f in os.listdir(src_dir) if f.endswith(".json"))[:max_files]
    for f in files:
        os.symlink(os.path.join(src_dir, f), os.path.join(dst_dir, f))
    os.makedirs(os.path.join(root, "data", "misc"))
    with open(os.path.join(root, "data", "misc", "service_to_replica_new.pickle"), "wb") as fh:
        pickle.dump({}, fh)
    os.makedirs(os.path.join(root, "results"))
    return root


def load_patched_v3(pydir):
    """Import algorithms.traceweaver_v3 with the debug block (lines 788-794) blanked."""
    path = os.path.join(pydir, "algorithms", "traceweaver_v3.py")
    with open(path) as fh:
        lines = fh.read().split("\n")
    a = next(i for i, l in enumerate(lines) if 'if ep1 == "client_ComposeReview"' in l)
    b = next(i for i, l in enumerate(lines) if 'print("t-statistic: "' in l)
    assert 780 < a < b < 800, (a, b)
    indent = lines[a][: len(lines[a]) - len(lines[a].lstrip())]
    for i in range(a, b + 1):
        lines[i] = indent + "pass"
    import algorithms  # noqa: F401  (namespace package from pydir)

    spec = importlib.util.spec_from_file_location("algorithms.traceweaver_v3", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["algorithms.traceweaver_v3"] = mod
    exec(compile("\n".join(lines), path, "exec"), mod.__dict__)
    return mod


class Recorder(object):
    """Collects per-service records while the reference runs."""

    def __init__(self):
        self.services = []
        self.cur = None

    # -- helpers -------------------------------------------------------------------------
    def _tuple_idx(self, spans):
        c = self.cur
        return [c["out_idx"][e][s.GetId()] for e, s in zip(c["out_eps"], spans[1:])]

    # -- wrappers ------------------------------------------------------------------------
    def install(self, v3mod):
        V3 = v3mod.TraceWeaverV3
        rec = self
        o_find = V3.FindAssignments
        o_topk = V3.FindTopKAssignments
        o_mis = V3.GetAssignmentsMIS
        o_p3 = V3.ComputeEpPairDistParams3
        o_p5 = V3.ComputeEpPairDistParams5

        def find(self_, method, process, in_parts, out_parts, parallel, hops, true_asg, graph, *a, **k):
            in_ep = list(in_parts.keys())[0]
            in_spans = in_parts[in_ep]
            out_eps = self_.GetOutEpsInOrder(out_parts, graph)
            c = {
                "process": process,
                "in_ep": in_ep,
                "out_eps": out_eps,
                "partition_key_order": list(out_parts.keys()),
                "n_in": len(in_spans),
                "in_start": np.array([s.start_mus for s in in_spans], dtype=np.int64),
                "in_dur": np.array([s.duration_mus for s in in_spans], dtype=np.int64),
                "in_idx": {s.GetId(): i for i, s in enumerate(in_spans)},
                "out_start": [np.array([s.start_mus for s in out_parts[e]], dtype=np.int64) for e in out_eps],
                "out_dur": [np.array([s.duration_mus for s in out_parts[e]], dtype=np.int64) for e in out_eps],
                "out_idx": {e: {s.GetId(): i for i, s in enumerate(out_parts[e])} for e in out_eps},
                "dag": np.array([[1 if graph.has_edge(p, q) else 0 for q in out_eps] for p in out_eps], dtype=np.uint8),
                "passes": [],
                "pre_leaves": [],
                "params3": [],
                "pass1_assign": None,
                "mixtures": None,
            }
            E = len(out_eps)
            c["true_parent"] = np.full((E, c["n_in"]), -1, dtype=np.int32)
            for ei, e in enumerate(out_eps):
                for sid, i in c["in_idx"].items():
                    oid = true_asg[e].get(sid)
                    if oid is not None and oid in c["out_idx"][e]:
                        c["true_parent"][ei, i] = c["out_idx"][e][oid]
            rec.cur = c
            t0 = time.time()
            ret = o_find(self_, method, process, in_parts, out_parts, parallel, hops, true_asg, graph, *a, **k)
            c["wall_s"] = time.time() - t0
            all_asg, all_topk, not_best, n_in, per_span_cand, cnt_unassigned = ret
            c["final_parent"] = rec._assign_to_idx(all_asg)
            ft = np.full((E, c["n_in"], TOPK), -1, dtype=np.int32)
            for ei, e in enumerate(out_eps):
                for sid, lst in all_topk[e].items():
                    for kk, oid in enumerate(lst):
                        ft[ei, c["in_idx"][sid], kk] = c["out_idx"][e][oid]
            c["final_topk"] = ft
            c["not_best_count"] = not_best
            c["cnt_unassigned"] = cnt_unassigned
            pc = np.zeros(c["n_in"], dtype=np.int64)
            for sid, v in per_span_cand.items():
                if sid in c["in_idx"]:
                    pc[c["in_idx"][sid]] = v
            c["per_span_candidates"] = pc
            c["windows"] = np.array(self_.span_windows, dtype=np.int32).reshape(-1, 2)
            rec.services.append(c)
            rec.cur = None
            return ret

        def topk(self_, in_eps, in_span, out_eps, out_parts, K, graph, id_x, preprocess_phase=False, count_candidates_phase=True):
            res = o_topk(self_, in_eps, in_span, out_eps, out_parts, K, graph, id_x, preprocess_phase, count_candidates_phase)
            c = rec.cur
            if c is None:
                return res
            if preprocess_phase:
                c["pre_leaves"].append(len(res))
                return res
            kind = "topk" if count_candidates_phase else "topk2"
            if kind == "topk" and id_x == 0:
                c["passes"].append({"topk": [], "topk2": [], "chosen": [], "mis_sizes": []})
            p = c["passes"][-1]
            p[kind].append([(float(sc), rec._tuple_idx(spans)) for sc, spans in res])
            return res

        def mis(self_, top_assignments):
            res = o_mis(self_, top_assignments)
            c = rec.cur
            if c is not None:
                p = c["passes"][-1]
                for ind, a in enumerate(res):
                    k = -1
                    for kk, (sc, spans) in enumerate(top_assignments[ind]):
                        if a is spans:
                            k = kk
                    p["chosen"].append(k)
                p["mis_sizes"].append(sum(len(t) for t in top_assignments))
            return res

        def p3(self_, in_parts, out_parts, out_eps, s, e, graph):
            o_p3(self_, in_parts, out_parts, out_eps, s, e, graph)
            c = rec.cur
            if c is not None:
                snap = {k: (float(v[0]), float(v[1])) for k, v in self_.services_times.items() if isinstance(v, tuple)}
                c["params3"].append((s, e, snap))

        def p5(self_, in_parts, out_parts, graph, all_asg, true_asg):
            # V3:1221-1222 sits inside the iteration loop: it runs after pass 1 *and* after pass 2.
            # Only the first call produces the mixtures that pass 2 scores with.
            c = rec.cur
            first = c is not None and c["pass1_assign"] is None
            if first:
                c["pass1_assign"] = rec._assign_to_idx(all_asg)
            o_p5(self_, in_parts, out_parts, graph, all_asg, true_asg)
            if first:
                mix = {}
                for k, v in self_.services_times.items():
                    if hasattr(v, "weights_"):
                        mix[k] = (
                            np.array(v.weights_, dtype=np.float64),
                            np.array(v.means_, dtype=np.float64).ravel(),
                            np.array(v.covariances_, dtype=np.float64).ravel(),
                            np.array(v.precisions_cholesky_, dtype=np.float64).ravel(),
                        )
                    elif isinstance(v, tuple):
                        mix[k] = (float(v[0]), float(v[1]))
                c["mixtures"] = mix

        # scikit-learn >= 1.6 allocates log_prob with X.dtype in _estimate_log_gaussian_prob, so the
        # reference's integer `np.array([t2 - t1])` (traceweaver_v1.py:125-126) gets its squared
        # distance and log(2*pi) truncated to int64.  The reference pins scikit-learn==1.5.1
        # (requirements.txt:22) where log_prob is float64; restore that behaviour.
        from sklearn import mixture as _mixture
        if not getattr(_mixture.GaussianMixture.score, "_tw_float_input", False):
            o_score = _mixture.GaussianMixture.score

            def score(self_, X, y=None):
                return o_score(self_, np.asarray(X, dtype=np.float64), y)

            score._tw_float_input = True
            _mixture.GaussianMixture.score = score

        V3.FindAssignments = find
        V3.FindTopKAssignments = topk
        V3.GetAssignmentsMIS = mis
        V3.ComputeEpPairDistParams3 = p3
        V3.ComputeEpPairDistParams5 = p5

    def _assign_to_idx(self, all_asg):
        c = self.cur
        E = len(c["out_eps"])
        out = np.full((E, c["n_in"]), -1, dtype=np.int32)
        for ei, e in enumerate(c["out_eps"]):
            for sid, oid in all_asg.get(e, {}).items():
                if oid in c["out_idx"][e]:
                    out[ei, c["in_idx"][sid]] = c["out_idx"][e][oid]
                elif oid == ("Skip", "Skip"):
                    out[ei, c["in_idx"][sid]] = -2
        return out


def edge_keys(c):
    """Scored edges in a fixed order: root(e) for DAG roots, primary (p,e), closing (e)."""
    return c["in_ep"], c["out_eps"]


def pack_service(dataset, c):
    E = len(c["out_eps"])
    n = c["n_in"]
    in_ep, out_eps = c["in_ep"], c["out_eps"]
    d = {
        "dataset": np.array(dataset),
        "process": np.array(c["process"]),
        "in_ep": np.array(in_ep),
        "out_eps": np.array(out_eps),
        "partition_key_order": np.array(c["partition_key_order"]),
        "seed": np.array(SEED),
        "in_start": c["in_start"],
        "in_dur": c["in_dur"],
        "out_off": np.cumsum([0] + [len(a) for a in c["out_start"]]).astype(np.int64),
        "out_start": np.concatenate(c["out_start"]),
        "out_dur": np.concatenate(c["out_dur"]),
        "dag": c["dag"],
        "true_parent": c["true_parent"],
        "windows": c["windows"],
        "pre_leaves": np.array(c["pre_leaves"], dtype=np.int64),
        "final_parent": c["final_parent"],
        "final_topk": c["final_topk"],
        "not_best_count": np.array(c["not_best_count"]),
        "cnt_unassigned": np.array(c["cnt_unassigned"]),
        "per_span_candidates": c["per_span_candidates"],
        "ref_wall_s": np.array(c["wall_s"]),
        "n_passes": np.array(len(c["passes"])),
    }
    if c["pass1_assign"] is not None:
        d["pass1_parent"] = c["pass1_assign"]
    # pass-1 Gaussian parameters: slots [root e | prim p*E+e | close e]
    nslot = E + E * E + E
    nb = len(c["params3"])
    par = np.full((nb, nslot, 2), np.nan)
    for b, (s, e_, snap) in enumerate(c["params3"]):
        for (k1, k2), (m, sd) in snap.items():
            if k1 == in_ep and k2 in out_eps:
                par[b, out_eps.index(k2)] = (m, sd)
            elif k2 == in_ep and k1 in out_eps:
                par[b, E + E * E + out_eps.index(k1)] = (m, sd)
            elif k1 in out_eps and k2 in out_eps:
                par[b, E + out_eps.index(k1) * E + out_eps.index(k2)] = (m, sd)
    d["params3"] = par
    d["params3_range"] = np.array([(s, e_) for s, e_, _ in c["params3"]], dtype=np.int64).reshape(-1, 2)
    # mixtures after pass 1: per slot up to 5 components (weights, means, covariances, prec_chol)
    if c["mixtures"] is not None:
        mixn = np.zeros(nslot, dtype=np.int32)
        mixp = np.zeros((nslot, 5, 4))
        for (k1, k2), v in c["mixtures"].items():
            if k1 == in_ep and k2 in out_eps:
                slot = out_eps.index(k2)
            elif k2 == in_ep and k1 in out_eps:
                slot = E + E * E + out_eps.index(k1)
            elif k1 in out_eps and k2 in out_eps:
                slot = E + out_eps.index(k1) * E + out_eps.index(k2)
            else:
                continue
            if isinstance(v, tuple) and len(v) == 4:
                w, mu, cov, pc = v
                mixn[slot] = len(w)
                mixp[slot, : len(w), 0] = w
                mixp[slot, : len(w), 1] = mu
                mixp[slot, : len(w), 2] = cov
                mixp[slot, : len(w), 3] = pc
            # tuple (mean, std) entries are BuildDistributions leftovers / (0,0) fallbacks
            elif isinstance(v, tuple) and len(v) == 2 and not np.isnan(par[0, slot, 0]) and v == (0, 0):
                mixn[slot] = -1
        d["mix_n"] = mixn
        d["mix_p"] = mixp
    for pi, p in enumerate(c["passes"]):
        for kind in ("topk", "topk2"):
            cnt = np.zeros(n, dtype=np.int32)
            idx = np.full((n, TOPK, E), -1, dtype=np.int32)
            sc = np.full((n, TOPK), np.nan)
            for i, lst in enumerate(p[kind]):
                cnt[i] = len(lst)
                for k, (s, tup) in enumerate(lst):
                    idx[i, k] = tup
                    sc[i, k] = s
            d["p%d_%s_n" % (pi, kind)] = cnt
            d["p%d_%s_idx" % (pi, kind)] = idx
            d["p%d_%s_score" % (pi, kind)] = sc
        d["p%d_chosen" % pi] = np.array(p["chosen"], dtype=np.int32)
        d["p%d_mis_sizes" % pi] = np.array(p["mis_sizes"], dtype=np.int32)
    return d


def run_dataset(name, rel_dir, fix):
    root = make_scratch_root(rel_dir)
    pydir = os.path.join(root, "src", "trace_reconstructor", "ports", "python")
    saved_path, saved_argv, saved_mods = list(sys.path), list(sys.argv), set(sys.modules)
    try:
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = load_patched_v3(pydir)
        rec = Recorder()
        rec.install(v3mod)
        sys.argv = [
            "executor.py", "--relative_path", rel_dir, "--compressed", "0", "--cache_rate", "0",
            "--fix", str(fix), "--test_name", name, "--load_level", "100", "--compress_factor", "1",
            "--repeat_factor", "1", "--execute_parallel", "0",
            "--results_directory", os.path.join(root, "results") + "/", "--clear_cache", "1",
            "--predictor_indices", "10",
        ]
        np.random.seed(SEED)
        devnull = open(os.devnull, "w")
        saved_stdout = sys.stdout
        sys.stdout = devnull
        try:
            runpy.run_path(os.path.join(pydir, "executor.py"), run_name="__main__")
        finally:
            sys.stdout = saved_stdout
        with open(os.path.join(root, "results", [f for f in os.listdir(os.path.join(root, "results")) if f.startswith("accuracy_")][0]), "rb") as fh:
            acc = pickle.load(fh)
        for c in rec.services:
            d = pack_service(name, c)
            d["e2e_accuracy"] = np.array(acc["MaxScoreBatchSubsetWithSkips"])
            d["e2e_topk_accuracy"] = np.array(acc["MaxScoreBatchSubsetWithSkipsTopK"])
            out = os.path.join(GOLDEN_DIR, "ref_%s__%s.npz" % (name, c["process"]))
            np.savez_compressed(out, **d)
            print("wrote", out, "n_in", c["n_in"], "E", len(c["out_eps"]), "wall %.1fs" % c["wall_s"], flush=True)
        print(name, "e2e accuracy", acc, flush=True)
    finally:
        sys.path[:] = saved_path
        sys.argv[:] = saved_argv
        for m in set(sys.modules) - saved_mods:
            del sys.modules[m]
        shutil.rmtree(root, ignore_errors=True)


def accuracy_band(names, seeds):
    """The reference's own run-to-run spread (SURVEY.md hazard H9: its model-selection fits draw from numpy's global RNG):
    the unmodified reference with several seeds per corpus -> tests/golden/ref_accuracy_band.json
    {corpus: {"seeds": [...], "e2e": [...], "e2e_topk": [...], "per_service": {service: [...]}}}."""
    import json

    global SEED
    path = os.path.join(GOLDEN_DIR, "ref_accuracy_band.json")
    band = json.load(open(path)) if os.path.exists(path) else {}
    saved_dir = GOLDEN_DIR
    for name, rel, fix in DATASETS:
        if name not in names:
            continue
        rec = band.setdefault(name, {"seeds": [], "e2e": [], "e2e_topk": [], "per_service": {}})
        for seed in seeds:
            if seed in rec["seeds"]:
                continue
            SEED = seed
            tmp = tempfile.mkdtemp(prefix="twband_")
            globals()["GOLDEN_DIR"] = tmp
            try:
                run_dataset(name, rel, fix)
                for f in sorted(os.listdir(tmp)):
                    d = np.load(os.path.join(tmp, f))
                    acc = float(np.all(d["final_parent"] == d["true_parent"], axis=0).mean())
                    rec["per_service"].setdefault(str(d["process"]), []).append(acc)
                    e2e, e2ek = float(d["e2e_accuracy"]), float(d["e2e_topk_accuracy"])
                rec["seeds"].append(seed); rec["e2e"].append(e2e); rec["e2e_topk"].append(e2ek)
            finally:
                globals()["GOLDEN_DIR"] = saved_dir
                shutil.rmtree(tmp, ignore_errors=True)
            with open(path, "w") as fh:
                json.dump(band, fh, indent=1, sort_keys=True)
            print("band", name, "seed", seed, "e2e", e2e, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--band", nargs="*", default=None, help="corpora to run with several seeds (accuracy only)")
    ap.add_argument("--seeds", nargs="*", type=int, default=[10, 1, 2, 3, 4])
    args = ap.parse_args()
    if args.band is not None:
        accuracy_band(args.band, args.seeds)
        return
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, rel, fix in DATASETS:
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        run_dataset(name, rel, fix)
        print("%s done in %.0fs" % (name, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
