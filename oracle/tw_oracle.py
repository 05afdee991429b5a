"""ctypes front-end of the CPU oracle (oracle/tw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker / CPU baseline.  The product package (traceweaver_amd/) must never import this.

`run_service()` restates TraceWeaverV3.FindAssignments (traceweaver_v3.py:1087-1229, no-skip mode):
windows -> per-block Gaussian parameters -> pass 1 -> gap samples -> mixture refit -> pass 2.
The mixture refit (traceweaver_v3.py:764-786) uses scikit-learn exactly as the reference does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtw_oracle.so")
TOPK = 5
MAX_COMP = 5


def build(force=False):
    src = os.path.join(_HERE, "tw_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


class _Service(ctypes.Structure):
    _fields_ = [
        ("n_in", ctypes.c_int32), ("E", ctypes.c_int32),
        ("in_start", ctypes.c_void_p), ("in_end", ctypes.c_void_p),
        ("out_off", ctypes.c_void_p), ("out_start", ctypes.c_void_p), ("out_end", ctypes.c_void_p),
        ("dag", ctypes.c_void_p), ("key_rank", ctypes.c_void_p),
        ("batch_size", ctypes.c_int32), ("batch_size_mis", ctypes.c_int32), ("topk", ctypes.c_int32),
        ("time_scale", ctypes.c_double), ("float_time", ctypes.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for name in ("two_log", "two_exp", "two_log1p"):
            getattr(_lib, name).restype = ctypes.c_double
            getattr(_lib, name).argtypes = [ctypes.c_double]
        _lib.two_windows.restype = ctypes.c_int
        _lib.two_gauss_params.restype = ctypes.c_int
        _lib.two_run_pass.restype = ctypes.c_int
        _lib.two_gaps.restype = ctypes.c_int
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


class Service(object):
    """SoA view of one service: endpoints in topological order, spans sorted by (start, end)."""

    def __init__(self, in_start, in_dur, out_off, out_start, out_dur, dag, key_rank=None,
                 batch_size=100, batch_size_mis=30, topk=TOPK, time_scale=None):
        # time_scale (a power of two): the int64 inputs are load-scaled timestamps in units of time_scale microseconds
        self.time_scale, self.float_time = (1.0, 0) if time_scale is None else (float(time_scale), 1)
        if np.asarray(in_start).dtype.kind == "f":
            # load-scaled inputs (helpers/transforms.py:10-40): float64 starts, int durations; ends = fl(start + dur)
            # as the reference forms them.  Handed to the C code as exact integers in units of 2^-k microseconds.
            fs = [np.asarray(in_start, dtype=np.float64), np.asarray(out_start, dtype=np.float64)]
            fe = [fs[0] + np.asarray(in_dur, dtype=np.float64), fs[1] + np.asarray(out_dur, dtype=np.float64)]
            k = exact_binary_exponent(fs + fe)
            self.time_scale, self.float_time = float(np.ldexp(1.0, -k)), 1
            self.in_start, self.out_start = (np.ldexp(a, k).astype(np.int64) for a in fs)
            self.in_end, self.out_end = (np.ldexp(a, k).astype(np.int64) for a in fe)
            for a, b in zip(fs + fe, (self.in_start, self.out_start, self.in_end, self.out_end)):
                assert np.array_equal(np.ldexp(b.astype(np.float64), -k), a)
        else:
            self.in_start = np.ascontiguousarray(in_start, dtype=np.int64)
            self.in_end = self.in_start + np.ascontiguousarray(in_dur, dtype=np.int64)
            self.out_start = np.ascontiguousarray(out_start, dtype=np.int64)
            self.out_end = self.out_start + np.ascontiguousarray(out_dur, dtype=np.int64)
        self.out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        self.E = len(self.out_off) - 1
        self.n_in = len(self.in_start)
        self.dag = np.ascontiguousarray(dag, dtype=np.uint8).reshape(self.E, self.E)
        self.key_rank = np.ascontiguousarray(
            np.arange(self.E) if key_rank is None else key_rank, dtype=np.int32)
        self.c = _Service(self.n_in, self.E, _p(self.in_start), _p(self.in_end), _p(self.out_off),
                          _p(self.out_start), _p(self.out_end), _p(self.dag), _p(self.key_rank),
                          batch_size, batch_size_mis, topk, self.time_scale, self.float_time)
        self.topk = topk
        self.batch_size = batch_size

    @property
    def nslot(self):
        return self.E * self.E + 2 * self.E

    def slot_root(self, e):
        return e

    def slot_prim(self, p, e):
        return self.E + p * self.E + e

    def slot_close(self, e):
        return self.E + self.E * self.E + e


def exact_binary_exponent(arrays):
    """Smallest k >= 0 such that every value of the float64 arrays is an integer multiple of 2^-k."""
    k = 0
    for a in arrays:
        a = np.asarray(a, dtype=np.float64)
        a = a[a != 0]
        if len(a) == 0:
            continue
        m, e = np.frexp(a)                                   # a = m * 2^e, 0.5 <= |m| < 1
        mant = np.abs(np.ldexp(m, 53)).astype(np.int64)      # 53-bit integer mantissa
        tz = np.zeros(len(a), dtype=np.int64)
        low = mant & -mant                                   # lowest set bit
        tz = np.log2(low.astype(np.float64)).astype(np.int64)
        k = max(k, int(np.max(53 - e - tz)))
    return k


def windows(svc):
    n = svc.n_in
    end_flag = np.zeros(n, dtype=np.uint8)
    pre = np.zeros(n, dtype=np.int64)
    win = np.zeros((n + 2, 2), dtype=np.int32)
    nw = lib().two_windows(ctypes.byref(svc.c), _p(end_flag), _p(pre), _p(win), ctypes.c_int32(n + 2))
    if nw < 0:
        raise RuntimeError("two_windows failed: %d" % nw)
    return end_flag, pre, win[:nw].copy()


def gauss_params(svc):
    nb = (svc.n_in + svc.batch_size - 1) // svc.batch_size
    g = np.zeros((nb, svc.nslot, 2), dtype=np.float64)
    rc = lib().two_gauss_params(ctypes.byref(svc.c), _p(g), ctypes.c_int32(nb))
    if rc < 0:
        raise RuntimeError("two_gauss_params failed: %d" % rc)
    return g


def run_pass(svc, end_flag, gauss=None, mix_n=None, mix_p=None, forced=None):
    """mode 0 (gauss given) or mode 1 (mixtures: mix_n [nslot] int32, mix_p [nslot,5,3] w/mean/prec_chol)."""
    n, E, K = svc.n_in, svc.E, svc.topk
    o = {
        "topk_n": np.zeros(n, np.int32), "topk_idx": np.zeros((n, K, E), np.int32), "topk_score": np.zeros((n, K)),
        "topk2_n": np.zeros(n, np.int32), "topk2_idx": np.zeros((n, K, E), np.int32), "topk2_score": np.zeros((n, K)),
        "leaves": np.zeros(n, np.int64), "chosen": np.full(n, -1, np.int32), "parent": np.zeros((E, n), np.int32),
    }
    stats = np.zeros(5, np.int64)
    mode = 0 if gauss is not None else 1
    if mode == 1:
        mix_n = np.ascontiguousarray(mix_n, dtype=np.int32)
        mix_p = np.ascontiguousarray(mix_p, dtype=np.float64)
        assert mix_p.shape == (svc.nslot, MAX_COMP, 3)
    else:
        gauss = np.ascontiguousarray(gauss, dtype=np.float64)
    rc = lib().two_run_pass(ctypes.byref(svc.c), ctypes.c_int(mode), _p(gauss), _p(mix_n), _p(mix_p), _p(end_flag),
                            _p(o["topk_n"]), _p(o["topk_idx"]), _p(o["topk_score"]), _p(o["topk2_n"]),
                            _p(o["topk2_idx"]), _p(o["topk2_score"]), _p(o["leaves"]), _p(o["chosen"]),
                            _p(o["parent"]), _p(stats),
                            _p(np.ascontiguousarray(forced, dtype=np.int32)) if forced is not None else ctypes.c_void_p(0))
    if rc != 0:
        raise RuntimeError("two_run_pass failed: %d" % rc)
    o["not_best_count"], o["cnt_unassigned"], o["mwis_nodes"], o["n_windows"], o["budget_windows"] = (int(v) for v in stats)
    return o


def gaps(svc, parent):
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    g = np.zeros((svc.nslot, svc.n_in), dtype=np.float64)
    cnt = np.zeros(svc.nslot, dtype=np.int32)
    lib().two_gaps(ctypes.byref(svc.c), _p(parent), _p(g), _p(cnt))
    return [g[q, :cnt[q]].copy() if cnt[q] >= 0 else None for q in range(svc.nslot)]


def fit_mixture(durations, n_selected=None):
    """traceweaver_v3.py:764-786 for one edge: BIC over 1..min(5,#unique) diag fits (global numpy RNG),
    then a full-covariance refit with random_state=100.  Returns (n, [n,3] weight/mean/prec_chol)."""
    from sklearn import mixture

    d = np.asarray(durations, dtype=np.float64).reshape(-1, 1)
    if len(d) == 0:
        return 0, np.zeros((MAX_COMP, 3))
    if n_selected is None:
        max_n = min(len(np.unique(d)), 5)
        models, n_comps = [], []
        for n in range(1, max_n + 1):
            try:
                models.append(mixture.GaussianMixture(n_components=n, covariance_type="diag").fit(d))
                n_comps.append(n)
            except ValueError:
                continue
        n_selected = n_comps[int(np.argmin([m.bic(d) for m in models]))]
    g = mixture.GaussianMixture(n_components=n_selected, random_state=100).fit(d)
    p = np.zeros((MAX_COMP, 3))
    p[:n_selected, 0] = g.weights_
    p[:n_selected, 1] = g.means_.ravel()
    p[:n_selected, 2] = g.precisions_cholesky_.ravel()
    return n_selected, p


def run_service(svc, mixtures=None, seed=None):
    """Full two-pass restatement.  `mixtures` = (mix_n, mix_p) overrides the refit (used to compare
    pass 2 against the reference under identical fitted parameters, SURVEY.md hazard H9)."""
    if seed is not None:
        np.random.seed(seed)
    end_flag, pre, win = windows(svc)
    g = gauss_params(svc)
    p1 = run_pass(svc, end_flag, gauss=g)
    out = {"end_flag": end_flag, "pre_leaves": pre, "windows": win, "gauss": g, "pass1": p1}
    if mixtures is None:
        mix_n = np.zeros(svc.nslot, np.int32)
        mix_p = np.zeros((svc.nslot, MAX_COMP, 3))
        for q, d in enumerate(gaps(svc, p1["parent"])):
            if d is not None:
                mix_n[q], mix_p[q] = fit_mixture(d)
    else:
        mix_n, mix_p = mixtures
    out["mix_n"], out["mix_p"] = mix_n, mix_p
    out["pass2"] = run_pass(svc, end_flag, mix_n=mix_n, mix_p=mix_p)
    out["per_span_candidates"] = p1["leaves"] + out["pass2"]["leaves"]
    return out


def service_from_golden(d):
    return Service(d["in_start"], d["in_dur"], d["out_off"], d["out_start"], d["out_dur"], d["dag"],
                   key_rank=golden_key_rank(d))


def golden_key_rank(d):
    order = [str(x) for x in d["partition_key_order"]]
    return np.array([order.index(str(e)) for e in d["out_eps"]], dtype=np.int32)


# ------------------------------------------------------------------------------------------------------------------
# skip mode (exps/exp2): TallySkipSpans / WaterFill (traceweaver_v3.py:853-989), BuildDistributions (:108-172), one pass
# with skip spans (two_run_skip in tw_oracle.c).  Literal restatements on index arrays.
SKIP_BASE, SKIP_STRIDE = 1024, 128


class _Skip(ctypes.Structure):
    _fields_ = [("sorted_perm", ctypes.c_void_p), ("n_tw", ctypes.c_int32), ("tw_start", ctypes.c_void_p), ("pool", ctypes.c_void_p),
                ("dist", ctypes.c_void_p)]


def tally_skip_spans(svc, prior_windows=()):
    """traceweaver_v3.py:853-989.  Returns (time windows [(start, end, expected)] sorted by start, skip budget [E],
    skip spans per (endpoint, window) [E, n_windows]).  `prior_windows`: what self.time_windows already held (the
    reference never clears it between services, hazard H8)."""
    n, E, B = svc.n_in, svc.E, svc.c.batch_size_mis
    tw = list(prior_windows)
    window_start = int(svc.in_start[0])
    final_end = int(np.max(svc.in_end))
    for i in range(n):                                              # :976-987
        if i != 0 and i != n - 1 and i % B == 0:
            window_end = int(svc.in_end[i])
            tw.append((window_start, window_end, B))
            window_start = window_end
        elif i == n - 1:
            tw.append((window_start, final_end, B))
    budget = np.array([n - int(svc.out_off[e + 1] - svc.out_off[e]) for e in range(E)], dtype=np.int64)    # :972
    keys = sorted(tw, key=lambda x: x[0])                          # WaterFill's window_keys (stable)
    pool = np.zeros((E, len(tw)), dtype=np.int32)
    for e in range(E):                                              # TackleMismatch :918-962
        st = svc.out_start[svc.out_off[e]:svc.out_off[e + 1]]
        counts = {}
        for (a, b, _) in tw:
            counts[(a, b)] = int(((st > a) & (st <= b)).sum())
        alloc = {k[:2]: 0 for k in tw}
        skip_budget = int(budget[e])
        if skip_budget > 0:                                         # WaterFill :862-916
            nw = len(counts)                                        # len(window_diffs): distinct (start, end) keys
            index_to_key = {i: k[:2] for i, k in enumerate(keys)}
            existing = np.zeros(nw)
            expected = np.zeros(nw)
            for i in range(nw):
                existing[i] = counts[index_to_key[i]]
                expected[i] = keys[i][2]
            order = np.argsort(existing)[::-1]
            sorted_existing = existing[order]
            alloc_v = np.zeros(nw)
            lam, total_remaining = 0, 0
            for i in range(nw):
                lam = (skip_budget + np.sum(sorted_existing[:i + 1])) // (i + 1)
                total_remaining = (skip_budget + np.sum(sorted_existing[:i + 1])) % (i + 1)
                if lam <= sorted_existing[i]:
                    break
            remaining = 0
            for i in range(nw):
                want = max(lam - sorted_existing[i], 0)
                give = min(want, expected[i] - sorted_existing[i])
                remaining += want - give
                alloc_v[order[i]] = give
            total_remaining += remaining
            while total_remaining > 0:
                no_change = True
                for i in reversed(range(nw)):
                    if total_remaining > 0 and alloc_v[order[i]] < (expected[i] - sorted_existing[i]):
                        alloc_v[order[i]] += 1
                        no_change = False
                        total_remaining -= 1
                if no_change:
                    break
            for i in range(nw):
                alloc[index_to_key[i]] = alloc_v[i]
        for w, k in enumerate(keys):
            pool[e, w] = max(int(alloc[k[:2]]), 0)       # range(int(negative)) is empty (:950)
    return keys, budget, pool


def build_distributions(svc):
    """traceweaver_v3.py:108-172: merged time-ordered sweep; every span looks back (no further than the longest request)
    for its nearest qualifying predecessor and contributes one delay sample to the pair (predecessor's endpoint, own
    endpoint).  Returns [(E+1), (E+1), 2] (np.mean, np.std), NaN where no sample; index 0 = the incoming endpoint."""
    E, n = svc.E, svc.n_in
    rows = [(int(svc.in_start[i]), int(svc.in_end[i] - svc.in_start[i]), 0, True) for i in range(n)]
    for e in range(E):
        for x in range(int(svc.out_off[e]), int(svc.out_off[e + 1])):
            rows.append((int(svc.out_start[x]), int(svc.out_end[x] - svc.out_start[x]), 1 + e, False))
    rows.sort(key=lambda r: r[0])                                   # list.sort is stable: ties keep [in, ep 0, ep 1, ...] order
    large = max(int(svc.in_end[i] - svc.in_start[i]) for i in range(n))
    samples = {}
    for i, (st, du, ep, server) in enumerate(rows):
        if not server:                                              # client span :125-150
            par, ptype = None, None
            for j in range(i - 1, -1, -1):
                p = rows[j]
                if (st + du) - p[0] > large:
                    break
                if p[3]:
                    par, ptype = p, "server"
                    break
                if (not p[3]) and p[0] + p[1] < st and p[2] < ep:   # out_ep_order == endpoint index (topological order)
                    par, ptype = p, "client"
                    break
            if par is not None:
                samples.setdefault((par[2], ep), []).append(st - par[0] if ptype == "server" else st - (par[0] + par[1]))
        else:                                                       # server span :152-169
            par = None
            for j in range(i - 1, -1, -1):
                p = rows[j]
                if (st + du) - p[0] > large:
                    break
                if (not p[3]) and p[0] + p[1] < st + du:
                    par = p
                    break
            if par is not None:
                samples.setdefault((par[2], ep), []).append((st + du) - (par[0] + par[1]))
            samples.setdefault((ep, ep), []).append(du)
    tab = np.full((E + 1, E + 1, 2), np.nan)
    for (a, b), v in samples.items():
        tab[a, b] = (np.mean(v), np.std(v))
    return tab, large


def run_skip(svc, end_flag, keys, pool, dist):
    """One pass with skip spans (two_run_skip).  Candidate indices: >= 0 a span (position in the list as handed over),
    <= -SKIP_BASE a skip span -(SKIP_BASE + window * SKIP_STRIDE + position in the window's pool); parent: -2 skip."""
    n, E, K = svc.n_in, svc.E, svc.topk
    perm = np.concatenate([np.argsort(svc.out_start[svc.out_off[e]:svc.out_off[e + 1]], kind="stable") for e in range(E)]).astype(np.int32)
    tw_start = np.array([k[0] for k in keys], dtype=np.int64)
    pool = np.ascontiguousarray(pool, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    sk = _Skip(_p(perm), len(keys), _p(tw_start), _p(pool), _p(dist))
    o = {"topk_n": np.zeros(n, np.int32), "topk_idx": np.zeros((n, K, E), np.int32), "topk_score": np.zeros((n, K)),
         "topk2_n": np.zeros(n, np.int32), "topk2_idx": np.zeros((n, K, E), np.int32), "topk2_score": np.zeros((n, K)),
         "leaves": np.zeros(n, np.int64), "chosen": np.full(n, -1, np.int32), "parent": np.zeros((E, n), np.int32)}
    stats = np.zeros(5, np.int64)
    lib().two_run_skip.restype = ctypes.c_int
    rc = lib().two_run_skip(ctypes.byref(svc.c), ctypes.byref(sk), _p(end_flag), _p(o["topk_n"]), _p(o["topk_idx"]), _p(o["topk_score"]),
                            _p(o["topk2_n"]), _p(o["topk2_idx"]), _p(o["topk2_score"]), _p(o["leaves"]), _p(o["chosen"]), _p(o["parent"]), _p(stats))
    if rc != 0:
        raise RuntimeError("two_run_skip failed: %d" % rc)
    o["not_best_count"], o["cnt_unassigned"], o["mwis_nodes"], o["n_windows"], o["budget_windows"] = (int(v) for v in stats)
    return o


def decode_skip(idx):
    """(index with skips as -2 - pool position, time window or -1) from the raw candidate indices of run_skip."""
    idx = np.asarray(idx)
    skip = idx <= -SKIP_BASE
    code = np.where(skip, -idx - SKIP_BASE, 0)
    return np.where(skip, -2 - code % SKIP_STRIDE, idx), np.where(skip, code // SKIP_STRIDE, -1)
