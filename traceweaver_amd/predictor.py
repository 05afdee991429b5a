"""`TraceWeaverGPU`: drop-in for the reference's TraceWeaverV3 predictor (predictor index 10,
"MaxScoreBatchSubsetWithSkips", executor.py:899) on top of the HIP engine.

Same constructor and the same `FindAssignments` signature / 6-tuple as
algorithms/traceweaver_v3.py:1087-1229, so the executor can register

    ("MaxScoreBatchSubsetWithSkips", TraceWeaverGPU(all_spans, all_processes))

in its predictor table (executor.py:888-900) and everything downstream (accuracy, result pickles,
plots) runs unchanged -- see INTEGRATION.md.  Spans are only read through `.start_mus`,
`.duration_mus` and `.GetId()`; (trace_id, span_id) keys never reach the device.

Services whose endpoints all hold one outgoing span per incoming span run the two passes with the refit in between;
a service that is short of outgoing spans at some endpoint (the cache-hit experiments, exp2) runs the reference's
one-pass skip mode (traceweaver_v3.py:820-989,1155-1156; csrc/tw_skip.h): assignments may then be ('Skip', 'Skip').
"""
import threading
import warnings

import numpy as np

from . import gmm
from .engine import Engine, UnitArrays

NA = ("NA", "NA")
SKIP = ("Skip", "Skip")


def pack_unit(in_spans, out_span_partitions, out_eps, invocation_graph):
    """SoA form of one FindAssignments call.  `out_eps` = topological order (traceweaver_v1.py:37-39)."""
    E = len(out_eps)
    n_in = len(in_spans)
    # After the executor's load scaling (helpers/transforms.py:10-40, --compress_factor > 1) start_mus is a Python float;
    # such a unit goes to the engine as the exact integer image of those floats (traceweaver_amd/transforms.py).
    scaled = any(isinstance(s.start_mus, float) for s in in_spans[:1]) or any(
        isinstance(s.start_mus, float) for ep in out_eps for s in out_span_partitions[ep][:1])
    tdt = np.float64 if scaled else np.int64
    in_start = np.fromiter((s.start_mus for s in in_spans), dtype=tdt, count=n_in)
    in_dur = np.fromiter((s.duration_mus for s in in_spans), dtype=tdt, count=n_in)
    out_off = np.zeros(E + 1, dtype=np.int64)
    starts, ends = [], []
    for k, ep in enumerate(out_eps):
        spans = out_span_partitions[ep]
        st = np.fromiter((s.start_mus for s in spans), dtype=tdt, count=len(spans))
        du = np.fromiter((s.duration_mus for s in spans), dtype=tdt, count=len(spans))
        starts.append(st)
        ends.append(st + du)
        out_off[k + 1] = out_off[k] + len(spans)
    dag = np.zeros((E, E), dtype=np.uint8)
    for a, p in enumerate(out_eps):
        for b, q in enumerate(out_eps):
            if invocation_graph.has_edge(p, q):
                dag[a, b] = 1
    keys = list(out_span_partitions.keys())  # networkx in_edges() order = FindOrder's insertion order (executor.py:223-236)
    key_rank = np.array([keys.index(ep) for ep in out_eps], dtype=np.int32)
    if scaled:
        from .transforms import to_exact_units

        (i_s, i_e, o_s, o_e), scale = to_exact_units([in_start, in_start + in_dur, np.concatenate(starts), np.concatenate(ends)])
        return UnitArrays(i_s, i_e, out_off, o_s, o_e, dag, key_rank, time_scale=scale)
    return UnitArrays(in_start, in_start + in_dur, out_off, np.concatenate(starts), np.concatenate(ends), dag, key_rank)


def reference_fit_order(unit, partition_keys_rank):
    """Slots in the order ComputeEpPairDistParams5 fits them (traceweaver_v3.py:800-818): for every
    endpoint in partition-key order: root edge (if no predecessor), primary in-edges in in_edges() order,
    closing edge."""
    E = unit.E
    order = []
    for e in sorted(range(E), key=lambda k: partition_keys_rank[k]):
        preds = sorted([p for p in range(E) if unit.dag[p, e]], key=lambda k: partition_keys_rank[k])
        if not preds:
            order.append(e)
        for p in preds:
            two_hop = any(unit.dag[p, m] and unit.dag[m, e] for m in range(E) if m not in (p, e))
            if not two_hop:
                order.append(E + p * E + e)
        order.append(E + E * E + e)
    return order


class TraceWeaverGPU(object):
    def __init__(self, all_spans, all_processes, device=0, fit="device", replay_true_fit=True, lib_path=None):
        """The refit between the passes is the reference's procedure (traceweaver_v3.py:764-786) either way; its only
        randomness -- the k-means++ draws of the model-selection fits -- comes from numpy's global RNG, consumed in the
        reference's call order.  fit = "device" (default): the fits run on the GPU (csrc/tw_fit.h) on a tape of exactly
        the doubles scikit-learn would draw; fit = "sklearn": scikit-learn itself on the host (cross-check).  With
        `replay_true_fit` the draws of the fits the reference discards (on the *true* assignments, and the round it runs
        again after the second pass, traceweaver_v3.py:796-818,1221-1222) are consumed too -- a seeded run then
        reproduces a seeded reference run, and leaves the RNG where the reference leaves it for the next service."""
        self.all_spans = all_spans
        self.all_processes = all_processes
        if fit == "sklearn":   # the cross-check consumes numpy's RNG as the installed scikit-learn does: the draw schedule the
            import sklearn     # device refit replays (1, 3, 7, 10, 13 uniforms for 1..5 components) is that of scikit-learn >= 1.3

            import re

            ver = re.match(r"^(\d+)\.(\d+)", sklearn.__version__)   # ("1.6rc1", "1.8.dev0": only the leading numbers count; unparsable: no warning)
            if ver is not None and (int(ver.group(1)), int(ver.group(2))) < (1, 3):
                warnings.warn("scikit-learn %s draws its k-means++ seeds on another schedule than the device refit replays "
                              "(>= 1.3): fit='sklearn' and fit='device' will not reproduce each other" % sklearn.__version__)
        self.fit = fit
        self.replay_true_fit = replay_true_fit
        self._engine = Engine(device, lib_path=lib_path)
        self._lock = threading.Lock()   # one engine, one batch at a time: the executor's optional thread pool
                                        # (executor.py:1015-1023) may call one predictor instance from several threads
        self.last_timing = {}
        self.last_stats = {}            # counters of the last call, incl. budget_windows (selection not proven optimal)
        self._time_windows = []         # the reference never clears self.time_windows between services (SURVEY.md hazard H8)

    # ------------------------------------------------------------------------------------------
    def _replay_true_fit(self, unit, true_parent):
        """The reference fits every edge on the *true* assignments first and overwrites the result (traceweaver_v3.py:796-818);
        those fits still draw from numpy's global RNG, so a seeded run only reproduces a seeded reference run with them."""
        for q in reference_fit_order(unit, unit.key_rank):
            row = self._gap_row(unit, true_parent, q)
            if len(row):
                gmm.fit_edge_sklearn(row)

    def _advance_rng(self, unit, parent):
        """Consumes the uniforms the reference's fits on `parent` would draw (fit = "device": their results are never used).
        Per edge with samples: n = 1..min(5, #unique) fits of 1, 3, 7, 10, 13 draws; the refit has its own seeded RNG."""
        for q in reference_fit_order(unit, unit.key_rank):
            row = self._gap_row(unit, parent, q)
            if len(row):
                np.random.random_sample(Engine.FIT_ROW_DRAWS[min(len(np.unique(row)), gmm.MAX_COMP)])

    def _device_refit(self, unit, true_parent):
        eng = self._engine
        if self.replay_true_fit and true_parent is not None:
            self._advance_rng(unit, true_parent)
        max_n = eng.fit_rows()[0]
        off = np.zeros(unit.nslot, dtype=np.int64)
        pos = 0
        for q in reference_fit_order(unit, unit.key_rank):
            off[q] = pos
            pos += Engine.FIT_ROW_DRAWS[int(max_n[q])]
        eng.fit_mixtures(tape=np.random.random_sample(pos), slot_off=[off])

    def _host_refit(self, unit, gaps_pred, true_parent):
        order = reference_fit_order(unit, unit.key_rank)
        if self.replay_true_fit and true_parent is not None:
            self._replay_true_fit(unit, true_parent)
        mix_n = np.zeros(unit.nslot, dtype=np.int32)
        mix_p = np.zeros((unit.nslot, gmm.MAX_COMP, 3))
        for q in order:
            row = gaps_pred[q]
            row = row[~np.isnan(row)]
            mix_n[q], mix_p[q] = gmm.fit_edge_sklearn(row)
        return mix_n, mix_p

    @staticmethod
    def _gap_row(unit, parent, q):
        """Host-side gap samples for an arbitrary assignment (only used to replay the true-assignment fits)."""
        E, n = unit.E, unit.n_in
        ok = parent[0] >= 0
        us = 1.0 if unit.time_scale is None else unit.time_scale   # microseconds per timestamp unit (exact: a power of two)
        if q < E:
            e = q
            return (unit.out_start[unit.out_off[e] + parent[e][ok]] - unit.in_start[ok]).astype(np.float64) * us
        if q < E + E * E:
            p, e = divmod(q - E, E)
            return (unit.out_start[unit.out_off[e] + parent[e][ok]] - unit.out_end[unit.out_off[p] + parent[p][ok]]).astype(np.float64) * us
        e = q - E - E * E
        return (unit.in_end[ok] - unit.out_end[unit.out_off[e] + parent[e][ok]]).astype(np.float64) * us

    # ------------------------------------------------------------------------------------------
    def FindAssignments(self, method, process, in_span_partitions, out_span_partitions, parallel, instrumented_hops,
                        true_assignments, invocation_graph, true_skips=False, true_dist=False):
        with self._lock:
            return self._find_assignments(method, process, in_span_partitions, out_span_partitions, parallel, instrumented_hops,
                                          true_assignments, invocation_graph, true_skips, true_dist)

    def solve_arrays(self, unit, true_parent=None, process=""):
        """Both passes of one no-skip service on index arrays (what FindAssignments does between packing and unpacking;
        traceweaver_amd.executor --fit sklearn calls it service by service).  true_parent [E, n_in] (or None) is only used to
        replay the reference's discarded fits on the true assignments (fit="sklearn", replay_true_fit).  Returns the result
        dicts of pass 1 (leaves) and pass 2 (everything)."""
        if not self.replay_true_fit:
            true_parent = None
        eng = self._engine
        eng.load([unit])
        eng.run_pass1()
        t1 = eng.timing()
        r1 = eng.results(1, fields=("leaves",))[0]
        if self.fit == "device":
            self._device_refit(unit, true_parent)
        else:
            mix_n, mix_p = self._host_refit(unit, eng.gaps()[0], true_parent)
            eng.set_mixtures([mix_n], [mix_p])
        eng.run_pass2()
        self.last_timing = {"pass1": t1, "pass2": eng.timing()}
        r2 = eng.results(2)[0]
        if true_parent is not None:
            # ComputeEpPairDistParams5 also runs after the second iteration (`if iterations > 1` sits inside the loop,
            # traceweaver_v3.py:1221-1222): its results are never used, but it advances numpy's global RNG by one more round
            # of fits on the true and on the pass-2 assignments -- what the next service of a seeded run starts from
            if self.fit == "device":
                self._advance_rng(unit, true_parent)
                self._advance_rng(unit, r2["parent"])
            else:
                self._replay_true_fit(unit, true_parent)
                for q in reference_fit_order(unit, unit.key_rank):
                    row = self._gap_row(unit, r2["parent"], q)
                    if len(row):
                        gmm.fit_edge_sklearn(row)
        self.last_stats = {k: r2[k] for k in ("not_best_count", "cnt_unassigned", "n_windows", "repaired_windows", "budget_windows")}
        if r2["budget_windows"]:
            warnings.warn("%d window(s) of service %r hit the node budget of the exact selection search: the selection returned "
                          "for them is the best one found, not a proven optimum" % (r2["budget_windows"], process))
        return r1, r2

    def _find_assignments_skip(self, unit, in_ids, out_ids, out_eps, out_span_partitions, true_assignments):
        from . import skipmode

        if unit.time_scale is not None:
            raise NotImplementedError("skip mode on load-scaled (float) timestamps")
        eng = self._engine
        plan = skipmode.plan(eng, unit, prior_windows=self._time_windows)
        self._time_windows = list(plan.windows)
        eng.load([unit], skip=[plan])
        eng.run_pass1()
        self.last_timing = {"pass1": eng.timing()}
        r = eng.results(1)[0]
        n_in = unit.n_in
        pick = lambda ids, x: ids[x] if x >= 0 else (SKIP if x <= -2 else NA)
        all_assignments, all_topk_assignments = {}, {}
        for k, ep in enumerate(out_eps):
            par = r["parent"][k]
            all_assignments[ep] = {in_ids[i]: pick(out_ids[k], int(par[i])) for i in range(n_in)}
            tk = r["topk_idx"][:, k, :]
            all_topk_assignments[ep] = {in_ids[i]: [pick(out_ids[k], -2 if tk[j, i] < -1 else int(tk[j, i])) for j in range(int(r["topk_n"][i]))]
                                        for i in range(n_in)}
        per_span_candidates = {}
        for ep in out_span_partitions.keys():                     # traceweaver_v3.py:1096-1098
            for key in true_assignments[ep].keys():
                per_span_candidates[key] = 0
        for i, sid in enumerate(in_ids):
            per_span_candidates[sid] = int(r["leaves"][i])
        self.last_stats = {k: r[k] for k in ("not_best_count", "cnt_unassigned", "n_windows", "repaired_windows", "budget_windows")}
        return all_assignments, all_topk_assignments, r["not_best_count"], n_in, per_span_candidates, r["cnt_unassigned"]

    def _find_assignments(self, method, process, in_span_partitions, out_span_partitions, parallel, instrumented_hops,
                          true_assignments, invocation_graph, true_skips=False, true_dist=False):
        assert len(in_span_partitions) == 1                       # traceweaver_v3.py:1088
        if parallel or true_skips or true_dist or instrumented_hops:
            raise NotImplementedError("TraceWeaverGPU accelerates predictor 10 (MaxScoreBatchSubsetWithSkips) only")
        import networkx as nx

        in_ep, in_spans = list(in_span_partitions.items())[0]
        out_eps = list(nx.topological_sort(invocation_graph))     # traceweaver_v1.py:39
        n_in = len(in_spans)
        unit = pack_unit(in_spans, out_span_partitions, out_eps, invocation_graph)
        in_ids = [s.GetId() for s in in_spans]
        out_ids = [[s.GetId() for s in out_span_partitions[ep]] for ep in out_eps]
        if any(len(out_span_partitions[ep]) != n_in for ep in out_eps):   # traceweaver_v3.py:972,1155-1156: one pass with skip spans
            return self._find_assignments_skip(unit, in_ids, out_ids, out_eps, out_span_partitions, true_assignments)
        true_parent = None
        if self.replay_true_fit and true_assignments is not None:
            true_parent = np.full((unit.E, n_in), -1, dtype=np.int64)
            for k, ep in enumerate(out_eps):
                pos = {sid: j for j, sid in enumerate(out_ids[k])}
                for i, sid in enumerate(in_ids):
                    true_parent[k, i] = pos.get(true_assignments[ep].get(sid), -1)
            true_parent[:, (true_parent < 0).any(axis=0)] = -1

        r1, r2 = self.solve_arrays(unit, true_parent, process)
        if unit.time_scale is None:   # TallySkipSpans runs for every service (traceweaver_v3.py:1141): its windows stay in the predictor (hazard H8)
            from . import skipmode

            self._time_windows = self._time_windows + skipmode.time_windows(unit)

        all_assignments, all_topk_assignments = {}, {}
        for k, ep in enumerate(out_eps):
            ids = out_ids[k]
            par = r2["parent"][k]
            all_assignments[ep] = {in_ids[i]: (ids[par[i]] if par[i] >= 0 else NA) for i in range(n_in)}
            tk = r2["topk_idx"][:, k, :]
            all_topk_assignments[ep] = {in_ids[i]: [ids[tk[j, i]] for j in range(int(r2["topk_n"][i]))] for i in range(n_in)}
        per_span_candidates = {}
        for ep in out_span_partitions.keys():                     # traceweaver_v3.py:1096-1098
            for key in true_assignments[ep].keys():
                per_span_candidates[key] = 0
        leaves = r1["leaves"] + r2["leaves"]
        for i, sid in enumerate(in_ids):
            per_span_candidates[sid] = int(leaves[i])
        return (all_assignments, all_topk_assignments, r2["not_best_count"], n_in, per_span_candidates,
                r2["cnt_unassigned"])
