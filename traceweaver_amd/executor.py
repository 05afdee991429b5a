#!/usr/bin/env python3
"""Command-line surface of the reference's executor (executor.py:38-74) for predictor 10, end to end on the GPU:

    python -m traceweaver_amd.executor --relative_path data/hotel_reservation/hotel_load100/ --compressed 0 \\
        --cache_rate 0 --fix 2 --test_name hotel_test --load_level 100 --compress_factor 1 --repeat_factor 1 \\
        --execute_parallel 0 --results_directory results/ --clear_cache 1 --predictor_indices "10"

Same flags, same result files (executor.py:1235-1244): `accuracy_*` ({method: end-to-end %, method+"TopK": ...}),
`process_acc_*` ({(method, process_id): accuracy}), `confidence_scores_*` ({service: [accuracy, not_best_count,
requests]}), `bin_acc_*` ({method: [(percentile, accuracy, response time ms)]}, helpers/utils.py:187-214) and
`e2e_*` (true / predicted end-to-end traces; as lists of (trace id, span id) keys ordered by start time -- the
reference pickles its own Span objects there).  Everything between reading the directory and writing the files runs
in libtwgpu.so: native ingest (tw_corpus_*), both passes of every service in one batch, device refit, device
accuracy reductions.

Indices 3 (WAP5), 4 (FCFS) and 7 (vPath) run on the resident table as well (csrc/tw_baselines.h; identical to the
reference's classes, also on load-scaled units; skip-mode services and index 5, ArrivalOrder, as host code:
traceweaver_amd/baselines.py) and add their columns to the same files --
`exps/exp1`'s and `exps/exp5`'s "3,4,7,10" run as they are.
`--compress_factor N` (N > 1) applies the reference's load scaling (helpers/transforms.py:10-40, executor.py:1086-1097,
1146-1148) to every service before it is solved: per-service load factor max(1, ceil(N / #replicas)) with the replica
table read from data/misc/service_to_replica_new.pickle under the project root (executor.py:912), timestamps handed to
the engine as exact images of the reference's floats (traceweaver_amd/transforms.py).  `--repeat_factor` is accepted and,
as in the reference (repeat_change_spans never reads it), only shows up in the result file names.
`--cache_rate r` (r > 0, exps/exp2) injects cache hits into the service named "frontend" like the reference
(executor.py:1150-1152, helpers/transforms.py:153-238 = traceweaver_amd/skipmode.py) and solves it in skip mode on the device.
What it does not do (and says so instead of approximating): the other predictor indices (the older TraceWeaver
variants 0-2, 6, 8, 9), --cache_rate together with --compress_factor > 1, --parallel / --instrumented, tar archives
(--compressed 1).  For those keep the reference's executor and register the predictor.
The mixture refit between the passes is the reference's own procedure (traceweaver_v3.py:764-786) on the GPU (csrc/tw_fit.h).
`--fit device` (default) feeds its k-means++ seedings the doubles numpy's global RNG would hand scikit-learn at that point of
a reference run started with np.random.seed(--seed), service by service in the reference's order -- incl. the reseeding
np.random.seed(10) the reference performs when it reaches the service named "frontend" (create_cache_hits runs for it at
every cache rate, executor.py:1150-1152, helpers/transforms.py:155) -- so a seeded run reproduces a seeded reference run;
`--fit sklearn` runs scikit-learn itself on the host at the same place (cross-check); `--fit device-batch` solves all
services in one batch with draws from the engine's own MT19937(--seed): the same procedure on another random stream, as an
unseeded reference run is (SURVEY.md hazard H9).
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np

METHOD = "MaxScoreBatchSubsetWithSkips"   # predictors[10] (executor.py:888-900)
NA = ("NA", "NA")


def parse_args(argv=None):
    q = lambda s: str(s).strip("'")   # the reference declares these with type=ascii and strips the quotes again
    ap = argparse.ArgumentParser(description="Map incoming and outgoing spans at each service (MI355X engine).")
    ap.add_argument("--relative_path", type=q, default=None)
    ap.add_argument("--absolute_path", type=q, default=None)
    ap.add_argument("--compressed", type=int, default=0, choices=[0, 1])
    ap.add_argument("--load_level", type=int, default=0)
    ap.add_argument("--test_name", type=q, default="test")
    ap.add_argument("--parallel", type=int, default=0, choices=[0, 1])
    ap.add_argument("--instrumented", type=int, default=0, choices=[0, 1])
    ap.add_argument("--cache_rate", type=float, required=True)
    ap.add_argument("--fix", type=int, required=True)
    ap.add_argument("--repeat_factor", type=int, default=1)
    ap.add_argument("--compress_factor", type=float, default=1)
    ap.add_argument("--execute_parallel", type=int, default=1)
    ap.add_argument("--results_directory", type=q, required=True)
    ap.add_argument("--clear_cache", type=int, default=0)
    ap.add_argument("--predictor_indices", type=str, default="")
    # additions (defaults keep the reference's behaviour)
    ap.add_argument("--project_root", type=q, default=os.getcwd(), help="what --relative_path is relative to")
    ap.add_argument("--max_traces", type=int, default=1001, help="the reference's literal limit (executor.py:873); 0 = all")
    ap.add_argument("--replicas_file", type=q, default=None,
                    help="pickle {service: [replica ids]} for --compress_factor > 1 (default: data/misc/service_to_replica_new.pickle under --project_root, executor.py:912)")
    ap.add_argument("--span_cache", type=int, default=0, choices=[0, 1],
                    help="1: keep the parsed span table of the trace directory in it (tw_span_table.bin) and start from it next time; "
                         "--clear_cache 1 rebuilds it, as it does the reference's file-order cache.  Off by default: it writes into "
                         "the data directory")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--fit", default="device", choices=["device", "device-batch", "sklearn"],
                    help="mixture refit between the passes: the reference's procedure (traceweaver_v3.py:764-786) in every case.  'device' "
                         "(default) = on the GPU, service by service in the reference's order, k-means++ draws replayed from numpy's global "
                         "RNG seeded by --seed: reproduces a reference run started with np.random.seed(seed) (SURVEY.md hazard H9).  "
                         "'sklearn' = scikit-learn on the host at the same place (cross-check; the passes still run on the GPU).  "
                         "'device-batch' = all services in one batch, draws from the engine's own MT19937(seed): fastest")
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--allow_partial", type=int, default=0, choices=[0, 1],
                    help="go on when services of the corpus cannot be solved here (skip mode, < 2 requests, cyclic call order); they are "
                         "recorded under 'left_out' in the confidence_scores pickle.  Default: stop, because the accuracy files would "
                         "otherwise count those services as right")
    ap.add_argument("--engine_library", type=q, default=None, help=argparse.SUPPRESS)   # tests: host-emulation build
    args = ap.parse_args(argv)
    if args.relative_path is None and args.absolute_path is None:
        ap.error("At least one of --relative_path and --absolute_path is required")
    return args


BASELINES = {3: "WAP5", 4: "FCFS", 5: "ArrivalOrder", 7: "vPath"}   # predictors[3], [4], [5], [7] (executor.py:888-900), host code like the reference's


def requested(args):
    idx = [int(x) for x in args.predictor_indices.split(",") if x.strip() != ""]
    return idx if idx else [10]


def unsupported(args):
    problems = []
    bad = [i for i in requested(args) if i != 10 and i not in BASELINES]
    if bad:
        problems.append("--predictor_indices %s (index 10 runs on the GPU, %s as host baselines; not %s)"
                        % (args.predictor_indices, sorted(BASELINES), bad))
    if args.compressed:
        problems.append("--compressed 1")
    if args.parallel or args.instrumented:
        problems.append("--parallel / --instrumented")
    return problems


def scale_load(units, args, trace_id, corpus):
    """Load scaling of every unit (helpers/transforms.py:10-40) with the per-service factor of executor.py:1086-1097."""
    from . import transforms
    from .ingest import IngestedUnit

    path = args.replicas_file or os.path.join(args.project_root, "data/misc/service_to_replica_new.pickle")
    if not os.path.exists(path):
        raise SystemExit("--compress_factor %g needs the replica table %s ({service: [replica ids]}, executor.py:912)" % (args.compress_factor, path))
    with open(path, "rb") as f:
        replicas = pickle.load(f)
    out, factors, ranks = [], [], []
    for u in units:
        owner = u.service if u.service in replicas else corpus.loop_origin(u.service)   # "...-loop" stand-ins: executor.py:1092-1096
        if owner not in replicas:
            raise SystemExit("service %s is not in the replica table %s (the reference stops here too, executor.py:1098-1101)" % (u.service, path))
        factor = transforms.load_factor(args.compress_factor, len(replicas[owner]))
        print("Process: %s  replicas: %d  dynamic load factor: %d" % (u.service, len(replicas[owner]), factor))
        keys = [trace_id(t) for t in u.in_trace]
        s = transforms.compress_unit(u.arrays, u.true_parent, factor, trace_key=keys)
        rank = np.empty(len(keys), dtype=np.int32)
        rank[np.argsort(np.asarray(keys), kind="stable")] = np.arange(len(keys), dtype=np.int32)
        factors.append(factor)
        ranks.append(rank)
        out.append(IngestedUnit(s.arrays, s.true_parent, u.in_trace[s.in_perm], u.service, u.in_ep, u.out_eps, u.in_rows[s.in_perm],
                                [r[p] for r, p in zip(u.out_rows, s.out_perm)], u.process_id))
    return out, factors, ranks


def run(args):
    from . import baselines
    from .engine import Engine, EngineError
    from .ingest import REFERENCE_FIX, open_directory

    if args.fix not in REFERENCE_FIX:
        raise SystemExit("--fix %d is not one of the reference's values 0..5" % args.fix)
    first_span, surgery = REFERENCE_FIX[args.fix]
    directory = args.absolute_path.rstrip("\\") if args.absolute_path else os.path.join(args.project_root, args.relative_path)
    t0 = time.time()
    # the span table of the directory is kept next to the reference's own time_order_filenames.pickle, under the same rule:
    # trusted until --clear_cache 1 (executor.py:320-339; ingest.open_directory has the guard it adds)
    corpus, counts = open_directory(directory, lib_path=args.engine_library, first_span=first_span, max_traces=args.max_traces, fix=surgery,
                                    cache=bool(args.span_cache), clear_cache=bool(args.clear_cache))
    units, skipped, n_traces = corpus.units()
    print("Loaded %d traces, %d spans in %.2f s%s (%d files rejected, %d traces filtered); %d services to solve, left out: %s"
          % (counts["traces"], counts["spans"], time.time() - t0, " from the directory's span table cache" if corpus.from_cache else "",
             counts["files_rejected"], counts["traces_filtered"], len(units), skipped))
    if not units:
        raise SystemExit("no service of this corpus can be solved (see the counts above)")
    # `several_callers` is a skip the reference performs itself (executor.py:1126-1128).  A service in which some request
    # lacks a call to an endpoint (`skip_mode`) makes the reference raise KeyError in FindOrder (executor.py:242: GetGroundTruth,
    # helpers/utils.py:22-32, leaves that request out of true_assignments[ep]) -- the only skip-mode inputs it solves are
    # the ones --cache_rate makes after FindOrder; `too_small` / `cyclic_order` raise further down (traceweaver_v3.py:1119,
    # nx.topological_sort).  Leaving such services out silently would make the figures look better than they are: stop
    lost = {k: v for k, v in skipped.items() if k != "several_callers" and v > 0}
    if lost and not args.allow_partial:
        raise SystemExit("%d service(s) of this corpus cannot be solved by the native chain: %s.  Re-run with --allow_partial 1 to "
                         "solve the others (the result files then record what was left out)." % (sum(lost.values()), lost))
    table = corpus.span_table()
    names = corpus.trace_names()
    trace_id = lambda k: corpus.string(names[k])
    resident = None
    if args.compress_factor > 1:                                   # executor.py:1086-1097,1146-1148
        resident = units                                           # the device scales its own copy of the table (tw_scale_load)
        units, load_factors, trace_ranks = scale_load(units, args, trace_id, corpus)
    skip_units = set()
    if args.cache_rate > 0:                                        # executor.py:1150-1152: only the service named "frontend"
        from . import skipmode
        from .ingest import IngestedUnit

        for k, u in enumerate(units):
            if u.service != "frontend":
                continue
            if u.arrays.time_scale is not None:
                raise SystemExit("--cache_rate with --compress_factor > 1 is not supported (skip mode takes integer microseconds)")
            print("cache %: ", float(args.cache_rate) * 100)
            arr, truth, kept = skipmode.cache_hits(u.arrays, u.true_parent, args.cache_rate, in_trace=u.in_trace)
            units[k] = IngestedUnit(arr, truth, u.in_trace, u.service, u.in_ep, u.out_eps, u.in_rows,
                                    [r[kept] if e == 0 else r for e, r in enumerate(u.out_rows)], u.process_id)
            skip_units.add(k)
    key = lambda row: (trace_id(table["trace"][row]), corpus.string(table["span_id"][row]))
    seen = np.zeros(n_traces, dtype=bool)
    for u in units:
        seen[u.in_trace] = True
    total = int(seen.sum())
    roots = [x for x in np.flatnonzero(table["parent"] < 0) if seen[table["trace"][x]]]

    accuracy_overall, accuracy_per_process, confidence, traces_overall, bins = {}, {}, {}, {}, {}

    def record(method, parents, bad_flags, options=None):
        """accuracy_*, e2e_*, bin_acc_* entries of one method from its parent arrays (or, for WAP5, its option lists)
        and per-trace wrong-flags."""
        true_traces, pred_traces = {}, {}
        for k, (u, par) in enumerate(zip(units, parents)):
            for i in range(u.arrays.n_in):                       # helpers/utils.py:216-252
                tid = trace_id(u.in_trace[i])
                tt, pt = true_traces.setdefault(tid, []), pred_traces.setdefault(tid, [])
                for e in range(u.arrays.E):
                    tt.append(int(u.out_rows[e][u.true_parent[e, i]]) if u.true_parent[e, i] >= 0 else None)
                    if options is not None:
                        pt.extend([int(u.out_rows[e][x]) for x in options[k][e][i]] or [None])
                    else:
                        pt.append(int(u.out_rows[e][par[e, i]]) if par[e, i] >= 0 else None)
        order = lambda rows: [key(x) if x is not None else None for x in sorted(rows, key=lambda x: float("inf") if x is None else table["start"][x])]
        traces_overall[method] = [{t: order(v) for t, v in true_traces.items()}, {t: order(v) for t, v in pred_traces.items()}]
        for name, bad in bad_flags.items():
            accuracy_overall[name] = int((~bad.astype(bool) & seen).sum()) / total * 100
            # BinAccuracyByResponseTimes (helpers/utils.py:187-214): traces ordered by the duration of their root span
            rows = sorted((int(table["duration"][x]), trace_id(table["trace"][x]), int(bad[table["trace"][x]] == 0)) for x in roots)
            acc, prev_c, prev_n, csum = [], 0, 0, np.cumsum([c for _, _, c in rows])
            for b in range(10):
                j = int(len(rows) * (b + 1) / 10 - 1)
                c, n = int(csum[j]) - prev_c, (j + 1) - prev_n
                prev_c, prev_n = prev_c + c, prev_n + n
                acc.append(((b + 1) * 100 / 10, c / n, rows[j][0] / 1000.0))
            bins[name] = acc

    for index in requested(args):
        if index == 10:
            eng = Engine(args.device, lib_path=args.engine_library)
            t1 = time.time()
            plain = [k for k in range(len(units)) if k not in skip_units]
            per, res = [None] * len(units), [None] * len(units)
            flags = np.zeros((2, n_traces), dtype=np.uint8)
            if plain and args.fit in ("sklearn", "device"):   # the reference's RNG stream replayed service by service
                from . import skipmode
                from .predictor import TraceWeaverGPU

                eng.close()
                pred = TraceWeaverGPU({}, {}, device=args.device, fit=args.fit, lib_path=args.engine_library)
                np.random.seed(args.seed)
                for k, u in enumerate(units):
                    if u.service == "frontend":   # executor.py:1150-1152: create_cache_hits reseeds (and draws) at every cache rate
                        skipmode.cache_hit_draws(u.arrays.n_in, args.cache_rate)
                    if k in skip_units:
                        continue                  # one pass, no refit: no draws (traceweaver_v3.py:1155-1156,1221)
                    _, r2 = pred.solve_arrays(u.arrays, u.true_parent, u.service)
                    pred._engine.set_truth([u.true_parent], [u.in_trace], n_traces)
                    p_, _, f_ = pred._engine.evaluate(trace_flags=True)
                    flags = np.maximum(flags, f_)
                    per[k], res[k] = p_[0], r2
                eng = pred._engine
            elif plain and resident is not None:
                # load levels on the resident table: the spans go up once, as ingested; the host transform above only
                # serves the result files and the baselines (same permutations: tests/test_load_scaling.py)
                eng.load([resident[k].arrays for k in plain])
                eng.set_truth([resident[k].true_parent for k in plain], [resident[k].in_trace for k in plain], n_traces)
                t_scale = time.time()
                perms = eng.scale_load([load_factors[k] for k in plain], trace_rank=[trace_ranks[k] for k in plain])
                for k, (ip, _, ts) in zip(plain, perms):
                    assert ts == units[k].arrays.time_scale and np.array_equal(resident[k].in_trace[ip], units[k].in_trace)
                print("Load scaling on the device: %.1f ms" % ((time.time() - t_scale) * 1e3))
            elif plain:
                eng.load([units[k].arrays for k in plain])
                eng.set_truth([units[k].true_parent for k in plain], [units[k].in_trace for k in plain], n_traces)
            if plain and args.fit == "device-batch":
                try:
                    eng.run_pass1()
                except EngineError as ex:
                    if ex.code == -7:   # TW_ERR_NAN_PARAMS, SURVEY.md hazard H3 (e.g. media_load75: 1500 files, 1001 traces kept)
                        raise SystemExit("%s\nThe reference fails on this input too (scipy.stats.tstd of one batch mean is NaN, "
                                         "traceweaver_v3.py:611). Keep a multiple-of-100-plus-anything-but-1 number of traces, "
                                         "e.g. --max_traces 1000." % ex)
                    raise
                eng.fit_mixtures(seed=args.seed)
                eng.run_pass2()
                p_, _, f_ = eng.evaluate(trace_flags=True)
                r_ = eng.results(2, fields=("parent", "unit_stats"))
                flags = np.maximum(flags, f_)
                for k, a, b in zip(plain, p_, r_):
                    per[k], res[k] = a, b
            if skip_units:   # services short of outgoing spans: the reference's one-pass skip mode (traceweaver_v3.py:1155-1156)
                from . import skipmode

                sk = sorted(skip_units)
                plans, windows = [], []
                for k, u in enumerate(units):   # one predictor object solves the services in turn and never clears its time windows:
                    if k in skip_units:         # those of every earlier service, skip budget or not, are still in it (hazard H8)
                        plans.append(skipmode.plan(eng, u.arrays, prior_windows=windows))
                        windows = list(plans[-1].windows)
                    elif u.arrays.time_scale is None:
                        windows = windows + skipmode.time_windows(u.arrays)
                eng.load([units[k].arrays for k in sk], skip=plans)
                eng.set_truth([units[k].true_parent for k in sk], [units[k].in_trace for k in sk], n_traces)
                eng.run_pass1()
                p_, _, f_ = eng.evaluate(trace_flags=True)
                r_ = eng.results(1, fields=("parent", "unit_stats"))
                flags = np.maximum(flags, f_)
                for k, a, b in zip(sk, p_, r_):
                    per[k], res[k] = a, b
            unproven = sum(r["budget_windows"] for r in res)
            if unproven:
                print("WARNING: %d window(s) hit the node budget of the exact selection search: their selection is the best one found, "
                      "not a proven optimum (DESIGN.md section 6)" % unproven)
            print("--- %s seconds --- (%d services, both passes, refit, accuracy)" % (time.time() - t1, len(units)))
            eng.close()
            for u, ev, r in zip(units, per, res):
                print("Accuracy for service %s: %.3f%%\n" % (u.service, ev["accuracy"] * 100))
                print("Top K accuracy for service %s: %.3f%%\n" % (u.service, ev["topk_accuracy"] * 100))
                accuracy_per_process[(METHOD, u.process_id)] = ev["accuracy"]
                confidence[u.service] = [ev["accuracy"], r["not_best_count"], u.arrays.n_in]
            record(METHOD, [r["parent"] for r in res], {METHOD: flags[0], METHOD + "TopK": flags[1]})
        else:
            method = BASELINES[index]
            parents, bad, all_options = [], np.zeros(n_traces, dtype=np.uint8), []
            on_device = method in ("FCFS", "vPath", "WAP5") and not skip_units
            if on_device:   # on the resident table (csrc/tw_baselines.h); skip-mode services (unsorted lists) and ArrivalOrder: host
                eng_b = Engine(args.device, lib_path=args.engine_library)
                eng_b.load([u.arrays for u in units])
                eng_b.set_truth([u.true_parent for u in units])
                if method == "WAP5":   # one predictor object for the run: delay samples accumulate per callee name over the services
                    parents, _, all_options = eng_b.wap5([u.out_eps for u in units])
                else:
                    parents = eng_b.baseline(method)
                eng_b.close()
            else:
                wap5 = baselines.WAP5()                                # one instance for the run: its state leaks across services
                fn = {"FCFS": lambda u: baselines.fcfs(u.arrays), "ArrivalOrder": lambda u: baselines.arrival_order(u.arrays),
                      "vPath": lambda u: baselines.vpath(u.arrays, u.true_parent), "WAP5": None}[method]
                for u in units:
                    if method == "WAP5":
                        all_options.append(wap5.assign(u.arrays, u.out_eps))
                        parents.append(baselines.WAP5.parent(u.arrays, all_options[-1]))
                    else:
                        parents.append(fn(u))
            for u, par in zip(units, parents):
                ok = np.all(par == u.true_parent, axis=0)
                print("Accuracy for service %s: %.3f%%\n" % (u.service, ok.mean() * 100))
                accuracy_per_process[(method, u.process_id)] = float(ok.mean())
                bad[u.in_trace[~ok]] = 1
            record(method, parents, {method: bad}, options=all_options if method == "WAP5" else None)
    for k, v in accuracy_overall.items():
        print("End-to-end accuracy for method %s: %.3f%%" % (k, v))
    if lost:
        confidence["left_out"] = dict(lost)

    os.makedirs(os.path.dirname(args.results_directory) or ".", exist_ok=True)   # the name is used as a prefix, like the reference does
    suffix = "_%s_%s_%s_%s_%s.pickle" % (args.test_name, args.load_level, int(args.compress_factor), int(args.repeat_factor), args.cache_rate)
    for name, obj in (("bin_acc", bins), ("accuracy", accuracy_overall), ("e2e", traces_overall), ("confidence_scores", confidence),
                      ("process_acc", accuracy_per_process)):
        with open(args.results_directory + name + suffix, "wb") as f:       # executor.py:1235-1244: plain concatenation
            pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)
    corpus.close()
    return accuracy_overall, accuracy_per_process, confidence


def main(argv=None):
    # this is an application: it asks for the hardware queues the engine's class streams want before HIP initialises
    # (csrc/tw_engine.hip, tw_create) -- unless the user chose a value
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    args = parse_args(argv)
    problems = unsupported(args)
    if problems:
        sys.exit("traceweaver_amd.executor runs predictor 10 (%s, with or without load scaling) and the baselines 3, 4, 5, 7; not supported here: %s.\n"
                 "Use the reference's executor with TraceWeaverGPU registered in its predictor table (INTEGRATION.md section 2)."
                 % (METHOD, "; ".join(problems)))
    run(args)


if __name__ == "__main__":
    main()
