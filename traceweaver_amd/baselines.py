"""The reference's baseline predictors on index arrays (executor.py:888-900 indices 3, 4, 5, 7):

    FCFS          algorithms/fcfs.py:10-26          request i takes the i-th call of every endpoint
    ArrivalOrder  algorithms/arrival_order.py:13-65  first endpoint FCFS, later ones in the order the previous calls returned
    vPath         algorithms/vpath.py:48-89          event sweep: a call belongs to the request that was last seen active
    WAP5          algorithms/wap5.py:257-351         most likely preceding request under an exponential delay model (class WAP5 below)

They are host code in the reference (Python) and stay host code here (numpy); they exist so that
`traceweaver_amd.executor` can serve the baseline columns of the reference's experiments next to the accelerated
predictor.  Inputs: a `UnitArrays` (endpoints in topological order; `key_rank` gives the partition-key order the
reference iterates in).  Output: parent [E, n_in] int32 in the unit's endpoint order, -1 = ("NA", "NA").
"""
import numpy as np


def _times(unit):
    """(in_start, in_end, out_start, out_end, in_dur, num): the timestamps as the reference's classes see them -- int64
    microseconds, or for a load-scaled unit (traceweaver_amd/transforms.py) the float64 values they are exact images of;
    `num` turns one of them into the Python number the reference computes with; durations stay the integers they were."""
    scale = getattr(unit, "time_scale", None)
    if scale is None:
        return unit.in_start, unit.in_end, unit.out_start, unit.out_end, unit.in_end - unit.in_start, int
    f = lambda a: a.astype(np.float64) * scale
    in_dur = getattr(unit, "in_dur", None)
    if in_dur is None:
        in_dur = np.rint(f(unit.in_end - unit.in_start)).astype(np.int64)
    return f(unit.in_start), f(unit.in_end), f(unit.out_start), f(unit.out_end), in_dur, float


def _key_order(unit):
    return [int(k) for k in np.argsort(unit.key_rank, kind="stable")]   # endpoints as out_span_partitions.items() yields them


def fcfs(unit):
    n = unit.n_in
    parent = np.full((unit.E, n), -1, dtype=np.int32)
    for e in range(unit.E):
        m = int(unit.out_off[e + 1] - unit.out_off[e])
        parent[e, :min(n, m)] = np.arange(min(n, m), dtype=np.int32)
    return parent


def arrival_order(unit):
    n = unit.n_in
    parent = np.full((unit.E, n), -1, dtype=np.int32)
    eps = _key_order(unit)
    _, _, out_start, out_end, _, num = _times(unit)
    # GetOutEpsInOrder (helpers/utils.py:15-21): endpoints by the start of their first span (list.sort is stable)
    o_eps = sorted(eps, key=lambda e: num(out_start[unit.out_off[e]]))
    prev = None   # indices (into the previous endpoint's list) in the order they are handed out
    for pos, e in enumerate(o_eps):
        m = int(unit.out_off[e + 1] - unit.out_off[e])
        if pos == 0:
            order = np.arange(m)
        else:
            pe = o_eps[pos - 1]
            ends = out_end[unit.out_off[pe]:unit.out_off[pe + 1]][prev]
            sort_order = list(np.argsort(ends))                      # arrival_order.py:40 (numpy's default sort, as there)
            if len(prev) <= m:
                sort_order = sort_order[:m] + list(range(len(prev), m))
            else:
                sort_order = [x for x in sort_order if x < m]
            order = np.array(sort_order, dtype=np.int64)
        k = min(n, len(order))
        parent[e, :k] = order[:k]
        prev = order
    return parent


def vpath(unit, true_parent):
    """`true_parent` [E, n_in] is only used the way the reference uses trace ids: to find the request a returning
    call belongs to (vpath.py:42-46,81-84)."""
    n, E = unit.n_in, unit.E
    eps = _key_order(unit)
    in_start, in_end, out_start, out_end, _, _ = _times(unit)
    t, key, kind, idx, epv = [], [], [], [], []
    # construction order of vpath.py:52-62: incoming spans (request, response), then every endpoint's calls
    t += [in_start, in_end]; key += [np.full(n, 1), np.full(n, 4)]; kind += [np.zeros(n, int), np.ones(n, int)]
    idx += [np.arange(n), np.arange(n)]; epv += [np.full(n, -1), np.full(n, -1)]
    # (request and response of one span are adjacent in the reference's list; interleave below)
    seq = [np.arange(n) * 2, np.arange(n) * 2 + 1]
    base = 2 * n
    owner = []
    for e in eps:
        a, b = int(unit.out_off[e]), int(unit.out_off[e + 1])
        m = b - a
        t += [out_start[a:b], out_end[a:b]]; key += [np.full(m, 2), np.full(m, 3)]
        kind += [np.full(m, 2), np.full(m, 3)]; idx += [np.arange(m), np.arange(m)]; epv += [np.full(m, e), np.full(m, e)]
        seq += [base + np.arange(m) * 2, base + np.arange(m) * 2 + 1]
        base += 2 * m
        inv = np.full(m, -1, dtype=np.int64)                          # request whose true call this is (first one wins)
        tp = np.asarray(true_parent[e])
        ok = tp >= 0
        inv[tp[ok][::-1]] = np.flatnonzero(ok)[::-1]
        owner.append(inv)
    t, key, kind, idx, epv, seq = (np.concatenate(x) for x in (t, key, kind, idx, epv, seq))
    first = np.argsort(seq, kind="stable")                            # the reference's list order
    t, key, kind, idx, epv = t[first], key[first], kind[first], idx[first], epv[first]
    order = np.lexsort((key, t.astype(np.float64)))                   # stable sort by (float(time), sort_key), vpath.py:64
    parent = np.full((E, n), -1, dtype=np.int32)
    own = {e: owner[k] for k, e in enumerate(eps)}
    latest = -1
    for j in order:
        k = kind[j]
        if k == 0:
            latest = int(idx[j])
        elif k == 1:
            latest = -1
        elif k == 2:
            if latest >= 0:
                parent[epv[j], latest] = idx[j]
        else:
            p = int(own[int(epv[j])][idx[j]])
            if p >= 0:
                latest = p
    return parent


def accuracy(parent, true_parent):
    """helpers/utils.py:62-79."""
    return float(np.all(parent == true_parent, axis=0).mean())


class WAP5(object):
    """algorithms/wap5.py:324-351 (FindAssignments and what it calls: BuildDistributions :257-275, ScoreParents
    :282-316).  The reference keeps one instance for all services of a run, and two pieces of its state leak from
    one service to the next: the delay samples per callee *name* (`distribution_values`) and the picked-flags of
    spans; both are reproduced, so call `assign` service after service in the executor's order.

    Returns per unit a list-valued assignment like the reference's ({request: [calls]} per endpoint), as
    `options[e][i]` = list of outgoing-span indices (empty = [("NA", "NA")]); `parent()` applies the reference's
    accuracy rule (helpers/utils.py:68-73: a request with more than one option for an endpoint counts as wrong)."""

    MAGIC_DELAY = 4

    def __init__(self):
        self.distribution_values = {}

    def assign(self, unit, ep_names):
        import math
        import statistics

        n, E = unit.n_in, unit.E
        in_start, _, out_start, _, in_dur, num = _times(unit)
        large_delay = int(in_dur.max())
        options = [[[] for _ in range(n)] for _ in range(E)]
        for e in _key_order(unit):                                   # for out_ep in out_span_partitions.keys()
            a, b = int(unit.out_off[e]), int(unit.out_off[e + 1])
            m = b - a
            # spans = incoming + outgoing of this endpoint, stable sort by start (incoming first on ties)
            start = np.concatenate([in_start, out_start[a:b]])
            is_client = np.concatenate([np.zeros(n, bool), np.ones(m, bool)])
            idx = np.concatenate([np.arange(n), np.arange(m)])
            order = np.argsort(start, kind="stable")
            start, is_client, idx = start[order], is_client[order], idx[order]
            values = self.distribution_values.setdefault(ep_names[e], [])
            # BuildDistributions: delay to the nearest preceding incoming span, if within the longest request duration
            for i in np.flatnonzero(is_client):
                sent = num(start[i])
                for j in range(i - 1, -1, -1):
                    if sent - num(start[j]) > large_delay:
                        break
                    if not is_client[j]:
                        values.append(sent - num(start[j]))
                        break
            # ScoreParents
            mean = statistics.mean(values)                            # constant during the loop below
            limit = self.MAGIC_DELAY * mean
            logpdf = lambda t: -t / mean - math.log(mean)             # scipy.stats.expon.logpdf(t, scale=mean)
            picked = np.zeros(n, bool)                                # already_picked, reset for the spans of this call
            for i in np.flatnonzero(is_client):
                sent = num(start[i])
                cands = []
                for j in range(i - 1, -1, -1):
                    if sent - num(start[j]) > limit:
                        cands.append((-1, logpdf(limit)))             # ("Spontaneous", p)
                        break
                    if not is_client[j] and not picked[idx[j]]:
                        cands.append((int(idx[j]), logpdf(sent - num(start[j]))))
                        picked[idx[j]] = True
                cands.sort(key=lambda x: x[1])                        # stable, ascending: the last entry wins
                if cands and cands[-1][0] >= 0:
                    options[e][cands[-1][0]].append(int(idx[i]))
        return options

    @staticmethod
    def parent(unit, options):
        par = np.full((unit.E, unit.n_in), -1, dtype=np.int32)
        for e in range(unit.E):
            for i, opt in enumerate(options[e]):
                par[e, i] = opt[0] if len(opt) == 1 else (-1 if not opt else -2)   # -2: several options = wrong
        return par
