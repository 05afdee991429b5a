#!/usr/bin/env python3
"""Golden vectors of the reference's *skip mode* (exps/exp2: cache hits injected into the hotel `frontend` service,
helpers/transforms.py:153-238; one pass with skip spans, traceweaver_v3.py:820-989,1136-1158).

TEST INFRASTRUCTURE ONLY -- runs the unmodified reference from /root/reference in the build container (stand-ins as in
gen_golden.py) with `--cache_rate R` and records, for the `frontend` call of FindAssignments (the only service the
executor injects cache hits into, executor.py:1150-1152):

  * the inputs exactly as the predictor receives them -- after create_cache_hits the outgoing lists are no longer sorted
    (spans of a cache-hit request were moved earlier in place, transforms.py:169-176) and the reference bisects them as
    they are (traceweaver_v3.py:1115 runs before the sort of :968-971);
  * the pre-transform inputs, the cache-hit request indices and the true assignments (-2 = ('Skip','Skip'));
  * time windows, skip budget, water-filled skip counts per (endpoint, time window)      traceweaver_v3.py:853-989
  * the (mean, std) table of BuildDistributions                                          traceweaver_v3.py:108-172
  * span windows, per request the top-5 lists of both FindTopKAssignments calls (skip spans as -2 - pool position, with
    the time window they were drawn from), the MWIS picks, the final assignment and the counters.

    python oracle/refrun/gen_golden_skip.py [--rates 0.05 0.1 0.2 0.3] [--dataset hotel_load150]
"""
import argparse
import os
import pickle
import runpy
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

TOPK = 5
DATASETS = {"hotel_load150": ("data/hotel_reservation/hotel_load150/", 2), "hotel_load100": ("data/hotel_reservation/hotel_load100/", 2),
            "hotel_load50": ("data/hotel_reservation/hotel_load50/", 2)}


class SkipRecorder(object):
    def __init__(self):
        self.services = []
        self.cur = None
        self.pre = None

    def install(self, v3mod, trmod):
        V3 = v3mod.TraceWeaverV3
        rec = self
        o_find, o_topk, o_mis = V3.FindAssignments, V3.FindTopKAssignments, V3.GetAssignmentsMIS
        o_tally, o_build = V3.TallySkipSpans, V3.BuildDistributions
        o_cache = trmod.create_cache_hits

        def cache(true_asg, in_parts, out_parts, cache_rate, exponential=False):
            in_ep = list(in_parts.keys())[0]
            rec.pre = {
                "in_start": np.array([s.start_mus for s in in_parts[in_ep]], dtype=np.int64),
                "in_dur": np.array([s.duration_mus for s in in_parts[in_ep]], dtype=np.int64),
                "keys": list(out_parts.keys()),
                "out_start": {e: np.array([s.start_mus for s in out_parts[e]], dtype=np.int64) for e in out_parts},
                "out_dur": {e: np.array([s.duration_mus for s in out_parts[e]], dtype=np.int64) for e in out_parts},
                "out_ids": {e: [s.GetId() for s in out_parts[e]] for e in out_parts},
                "in_ids": [s.GetId() for s in in_parts[in_ep]],
                "true": {e: dict(true_asg[e]) for e in out_parts},
            }
            return o_cache(true_asg, in_parts, out_parts, cache_rate, exponential)

        def find(self_, method, process, in_parts, out_parts, parallel, hops, true_asg, graph, *a, **k):
            in_ep = list(in_parts.keys())[0]
            in_spans = in_parts[in_ep]
            out_eps = self_.GetOutEpsInOrder(out_parts, graph)
            c = {"process": process, "in_ep": in_ep, "out_eps": out_eps, "partition_key_order": list(out_parts.keys()), "n_in": len(in_spans),
                 "in_start": np.array([s.start_mus for s in in_spans], dtype=np.int64),
                 "in_dur": np.array([s.duration_mus for s in in_spans], dtype=np.int64),
                 "in_idx": {s.GetId(): i for i, s in enumerate(in_spans)},
                 "out_start": [np.array([s.start_mus for s in out_parts[e]], dtype=np.int64) for e in out_eps],
                 "out_dur": [np.array([s.duration_mus for s in out_parts[e]], dtype=np.int64) for e in out_eps],
                 "out_idx": {e: {s.GetId(): i for i, s in enumerate(out_parts[e])} for e in out_eps},
                 "dag": np.array([[1 if graph.has_edge(p, q) else 0 for q in out_eps] for p in out_eps], dtype=np.uint8),
                 "passes": [], "pre_leaves": [], "skip_code": {}, "pre": rec.pre if process == "frontend" else None,
                 "dynamism_at_entry": bool(self_.dynamism), "time_windows_at_entry": len(self_.time_windows)}
            rec.pre = None
            E = len(out_eps)
            c["true_parent"] = np.full((E, c["n_in"]), -1, dtype=np.int32)
            for ei, e in enumerate(out_eps):
                for sid, i in c["in_idx"].items():
                    oid = true_asg[e].get(sid)
                    if oid == ("Skip", "Skip"):
                        c["true_parent"][ei, i] = -2
                    elif oid is not None and oid in c["out_idx"][e]:
                        c["true_parent"][ei, i] = c["out_idx"][e][oid]
            rec.cur = c
            t0 = time.time()
            ret = o_find(self_, method, process, in_parts, out_parts, parallel, hops, true_asg, graph, *a, **k)
            c["wall_s"] = time.time() - t0
            all_asg, all_topk, not_best, n_in, per_span_cand, cnt_unassigned = ret
            fp = np.full((E, c["n_in"]), -1, dtype=np.int32)
            for ei, e in enumerate(out_eps):
                for sid, oid in all_asg.get(e, {}).items():
                    fp[ei, c["in_idx"][sid]] = -2 if oid == ("Skip", "Skip") else (c["out_idx"][e][oid] if oid in c["out_idx"][e] else -1)
            c["final_parent"] = fp
            ft = np.full((E, c["n_in"], TOPK), -1, dtype=np.int32)
            for ei, e in enumerate(out_eps):
                for sid, lst in all_topk[e].items():
                    for kk, oid in enumerate(lst):
                        ft[ei, c["in_idx"][sid], kk] = -2 if oid == ("Skip", "Skip") else c["out_idx"][e][oid]
            c["final_topk"] = ft
            c["not_best_count"], c["cnt_unassigned"] = not_best, cnt_unassigned
            pc = np.zeros(c["n_in"], dtype=np.int64)
            for sid, v in per_span_cand.items():
                if sid in c["in_idx"]:
                    pc[c["in_idx"][sid]] = v
            c["per_span_candidates"] = pc
            c["windows"] = np.array(self_.span_windows, dtype=np.int32).reshape(-1, 2)
            c["in_order_after"] = np.array([c["in_idx"][s.GetId()] for s in in_spans], dtype=np.int32)
            c["out_order_after"] = [np.array([c["out_idx"][e][s.GetId()] for s in out_parts[e]], dtype=np.int32) for e in out_eps]
            rec.services.append(c)
            rec.cur = None
            return ret

        def tally(self_, in_parts, out_parts, in_eps, out_eps, batch_size_mis):
            o_tally(self_, in_parts, out_parts, in_eps, out_eps, batch_size_mis)
            c = rec.cur
            if c is None:
                return
            tw = sorted(self_.time_windows, key=lambda x: x[0])
            c["time_windows"] = np.array([(a, b, n) for a, b, n in tw], dtype=np.int64).reshape(-1, 3)
            c["skip_budget"] = np.array([self_.overall_skip_budget[e] for e in out_eps], dtype=np.int64)
            cnt = np.zeros((len(out_eps), len(tw)), dtype=np.int64)
            for ei, e in enumerate(out_eps):
                for wi, (a, b, _) in enumerate(tw):
                    pool = self_.available_skips_per_window[e][(a, b)]
                    cnt[ei, wi] = len(pool)
                    for pos, (sp, _) in enumerate(pool):
                        c["skip_code"][sp.sid] = (wi, pos)
            c["skip_count"] = cnt

        def build(self_, process, in_parts, out_parts, in_eps, out_eps):
            o_build(self_, process, in_parts, out_parts, in_eps, out_eps)
            c = rec.cur
            if c is None:
                return
            names = [c["in_ep"]] + list(c["out_eps"])
            tab = np.full((len(names), len(names), 2), np.nan)
            for (k1, k2), v in self_.services_times.items():
                if k1 in names and k2 in names and isinstance(v, tuple):
                    tab[names.index(k1), names.index(k2)] = (float(v[0]), float(v[1]))
            c["dist_table"] = tab            # index 0 = the incoming endpoint, 1 + e = outgoing endpoint e
            c["large_delay"] = int(self_.large_delay)

        def tuple_codes(spans):
            c = rec.cur
            idx, win = [], []
            for e, s in zip(c["out_eps"], spans[1:]):
                if s.trace_id == "None":
                    wi, pos = c["skip_code"][s.sid]
                    idx.append(-2 - pos)
                    win.append(wi)
                else:
                    idx.append(c["out_idx"][e][s.GetId()])
                    win.append(-1)
            return idx, win

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
            c["passes"][-1][kind].append([(float(sc),) + tuple_codes(spans) for sc, spans in res])
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

        V3.FindAssignments, V3.FindTopKAssignments, V3.GetAssignmentsMIS = find, topk, mis
        V3.TallySkipSpans, V3.BuildDistributions = tally, build
        trmod.create_cache_hits = cache


def pack(name, rate, c):
    E, n = len(c["out_eps"]), c["n_in"]
    d = {"dataset": np.array(name), "cache_rate": np.array(rate), "process": np.array(c["process"]), "in_ep": np.array(c["in_ep"]),
         "out_eps": np.array(c["out_eps"]), "partition_key_order": np.array(c["partition_key_order"]),
         "in_start": c["in_start"], "in_dur": c["in_dur"],
         "out_off": np.cumsum([0] + [len(a) for a in c["out_start"]]).astype(np.int64),
         "out_start": np.concatenate(c["out_start"]), "out_dur": np.concatenate(c["out_dur"]), "dag": c["dag"],
         "true_parent": c["true_parent"], "windows": c["windows"], "pre_leaves": np.array(c["pre_leaves"], dtype=np.int64),
         "final_parent": c["final_parent"], "final_topk": c["final_topk"], "not_best_count": np.array(c["not_best_count"]),
         "cnt_unassigned": np.array(c["cnt_unassigned"]), "per_span_candidates": c["per_span_candidates"],
         "n_passes": np.array(len(c["passes"])), "time_windows": c["time_windows"], "skip_budget": c["skip_budget"],
         "skip_count": c["skip_count"], "dist_table": c["dist_table"], "large_delay": np.array(c["large_delay"]),
         "in_order_after": c["in_order_after"], "out_order_after": np.concatenate(c["out_order_after"]),
         "dynamism_at_entry": np.array(c["dynamism_at_entry"]), "time_windows_at_entry": np.array(c["time_windows_at_entry"]),
         "ref_wall_s": np.array(c["wall_s"])}
    if c["pre"] is not None:
        pre = c["pre"]
        d["pre_in_start"], d["pre_in_dur"] = pre["in_start"], pre["in_dur"]
        d["pre_out_off"] = np.cumsum([0] + [len(pre["out_start"][e]) for e in c["out_eps"]]).astype(np.int64)
        d["pre_out_start"] = np.concatenate([pre["out_start"][e] for e in c["out_eps"]])
        d["pre_out_dur"] = np.concatenate([pre["out_dur"][e] for e in c["out_eps"]])
        tp = np.full((E, n), -1, dtype=np.int32)
        for ei, e in enumerate(c["out_eps"]):
            pos = {sid: j for j, sid in enumerate(pre["out_ids"][e])}
            for i, sid in enumerate(pre["in_ids"]):
                tp[ei, i] = pos.get(pre["true"][e].get(sid), -1)
        d["pre_true_parent"] = tp
        d["pre_partition_key_order"] = np.array(pre["keys"])
    for pi, p in enumerate(c["passes"]):
        for kind in ("topk", "topk2"):
            cnt = np.zeros(n, dtype=np.int32)
            idx = np.full((n, TOPK, E), -1, dtype=np.int32)
            win = np.full((n, TOPK, E), -1, dtype=np.int32)
            sc = np.full((n, TOPK), np.nan)
            for i, lst in enumerate(p[kind]):
                cnt[i] = len(lst)
                for k, (s, tup, w) in enumerate(lst):
                    idx[i, k], win[i, k], sc[i, k] = tup, w, s
            d["p%d_%s_n" % (pi, kind)], d["p%d_%s_idx" % (pi, kind)] = cnt, idx
            d["p%d_%s_win" % (pi, kind)], d["p%d_%s_score" % (pi, kind)] = win, sc
        d["p%d_chosen" % pi] = np.array(p["chosen"], dtype=np.int32)
        d["p%d_mis_sizes" % pi] = np.array(p["mis_sizes"], dtype=np.int32)
    return d


def run(name, rate):
    rel_dir, fix = DATASETS[name]
    root = G.make_scratch_root(rel_dir)
    pydir = os.path.join(root, "src", "trace_reconstructor", "ports", "python")
    saved_path, saved_argv, saved_mods = list(sys.path), list(sys.argv), set(sys.modules)
    try:
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = G.load_patched_v3(pydir)
        import helpers.transforms as trmod

        rec = SkipRecorder()
        rec.install(v3mod, trmod)
        sys.argv = ["executor.py", "--relative_path", rel_dir, "--compressed", "0", "--cache_rate", str(rate), "--fix", str(fix),
                    "--test_name", name, "--load_level", "150", "--compress_factor", "1", "--repeat_factor", "1", "--execute_parallel", "0",
                    "--results_directory", os.path.join(root, "results") + "/", "--clear_cache", "1", "--predictor_indices", "10"]
        np.random.seed(G.SEED)
        saved_stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            runpy.run_path(os.path.join(pydir, "executor.py"), run_name="__main__")
        finally:
            sys.stdout = saved_stdout
        with open(os.path.join(root, "results", [f for f in os.listdir(os.path.join(root, "results")) if f.startswith("accuracy_")][0]), "rb") as fh:
            acc = pickle.load(fh)
        for c in rec.services:
            d = pack(name, rate, c)
            d["e2e_accuracy"] = np.array(acc["MaxScoreBatchSubsetWithSkips"])
            d["e2e_topk_accuracy"] = np.array(acc["MaxScoreBatchSubsetWithSkipsTopK"])
            out = os.path.join(G.GOLDEN_DIR, "refskip_%s_c%s__%s.npz" % (name, str(rate).replace(".", "p"), c["process"]))
            np.savez_compressed(out, **d)
            print("wrote", out, "n_in", c["n_in"], "passes", len(c["passes"]), "skip budget", c["skip_budget"].tolist(), "wall %.1fs" % c["wall_s"], flush=True)
        print(name, rate, "e2e accuracy", acc, flush=True)
    finally:
        sys.path[:] = saved_path
        sys.argv[:] = saved_argv
        for m in set(sys.modules) - saved_mods:
            del sys.modules[m]
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", nargs="*", type=float, default=[0.05, 0.1, 0.2, 0.3])
    ap.add_argument("--dataset", default="hotel_load150")
    args = ap.parse_args()
    for r in args.rates:
        t0 = time.time()
        run(args.dataset, r)
        print("rate %s done in %.0fs" % (r, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
