"""Skip mode (exps/exp2): what the reference does when an endpoint holds fewer outgoing spans than there are requests
-- cache hits, helpers/transforms.py:153-238 -- before and around the one pass the engine then runs (csrc/tw_skip.h).

  cache_hits          helpers/transforms.py:153-238 create_cache_hits on index arrays (the executor's --cache_rate)
  tally_skip_spans    traceweaver_v3.py:853-989 TallySkipSpans + WaterFill: 30-request time windows, skip budget, pools
  build_distributions traceweaver_v3.py:108-172: merged start-ordered sweep on the device (Engine.build_distributions),
                      np.mean / np.std per endpoint pair on the host
  plan                the SkipPlan Engine.load(..., skip=[...]) takes

Host code operates on a few dozen windows and (E + 1)^2 pairs; the per-span work is on the device.
"""
import numpy as np

from .engine import UnitArrays

BATCH_MIS = 30


class SkipPlan(object):
    def __init__(self, windows, budget, pool, dist, large_delay):
        self.windows = windows                      # [(start, end, expected)] in start order
        self.tw_start = np.array([w[0] for w in windows], dtype=np.int64)
        self.budget = budget                        # [E] n_in - n_out
        self.pool = pool                            # [E, n_windows] skip spans per (endpoint, window)
        self.dist = dist                            # [(E+1), (E+1), 2] mean, std; NaN = no sample
        self.large_delay = large_delay


def cache_hit_draws(n, cache_rate, seed=10):
    """The RNG side of create_cache_hits (helpers/transforms.py:155,197-203): numpy's global RNG is reseeded with 10 and
    int(cache_rate * n) exponentials (unused) and as many indices without replacement, weights exp(-0.001 * index), are drawn
    -- for the service named "frontend" at every cache rate (executor.py:1150-1152), also 0 (then only the reseeding
    happens).  Returns the requests that become cache hits."""
    np.random.seed(seed)                                                  # transforms.py:155
    size = int(cache_rate * n)
    np.random.exponential(scale=1 / 0.001, size=size)                     # drawn and not used (transforms.py:197)
    p = np.asarray(np.exp(-0.001 * np.arange(n))).astype("float64")
    p = p / np.sum(p)
    return np.random.choice(np.arange(n), size=size, replace=False, p=p)


def cache_hits(unit, true_parent, cache_rate, seed=10, in_trace=None):
    """create_cache_hits (helpers/transforms.py:153-238): a fraction `cache_rate` of the requests, drawn without
    replacement with weights exp(-0.001 * index) from numpy's global RNG seeded with 10, lose their call to the FIRST
    endpoint (topological order): that span is deleted, the request is shortened by its duration and the request's
    spans at the later endpoints move earlier by the same amount -- in place, so the later endpoints' lists are no
    longer sorted.  Returns (unit as the predictor receives it, true_parent with -2 = ('Skip','Skip'), which spans of
    the first endpoint are kept)."""
    n, E = unit.n_in, unit.E
    tp0 = np.asarray(true_parent)
    # adjust_spans (transforms.py:170-180) moves every span that shares the request's trace id; on index arrays that is the
    # request's own call per endpoint -- which needs every request to have exactly one (a corpus with repeated calls per
    # trace, or a request that already lacks a call, is not what this restatement covers: say so instead of moving the wrong span)
    if in_trace is not None and len(np.unique(in_trace)) != len(in_trace):
        raise ValueError("cache_hits: several requests of one trace at this service (create_cache_hits shifts by trace id)")
    if (tp0 < 0).any():
        raise ValueError("cache_hits: a request without a call to some endpoint (create_cache_hits assumes one call per request and endpoint)")
    hit = cache_hit_draws(n, cache_rate, seed)
    hit_mask = np.zeros(n, dtype=bool)
    hit_mask[hit] = True
    in_end = unit.in_end.copy()
    starts = [unit.out_start[unit.out_off[e]:unit.out_off[e + 1]].copy() for e in range(E)]
    ends = [unit.out_end[unit.out_off[e]:unit.out_off[e + 1]].copy() for e in range(E)]
    tp = np.array(true_parent, dtype=np.int32).copy()
    keep0 = np.ones(len(starts[0]), dtype=bool)
    for i in np.flatnonzero(hit_mask):                                    # requests in order (transforms.py:207)
        x = int(tp[0, i])
        d = int(ends[0][x] - starts[0][x])
        in_end[i] -= d                                                    # adjust_spans: the request ...
        for e in range(1, E):                                             # ... and its spans at the later endpoints
            y = int(tp[e, i])
            starts[e][y] -= d
            ends[e][y] -= d
        keep0[x] = False
        tp[0, i] = -2
    new_index = np.cumsum(keep0) - 1                                      # positions after the deletions
    live = tp[0] >= 0
    tp[0, live] = new_index[tp[0, live]]
    starts[0], ends[0] = starts[0][keep0], ends[0][keep0]
    off = np.concatenate([[0], np.cumsum([len(s) for s in starts])]).astype(np.int64)
    out = UnitArrays(unit.in_start, in_end, off, np.concatenate(starts), np.concatenate(ends), unit.dag, unit.key_rank)
    return out, tp, keep0


def time_windows(unit):
    """The 30-request time windows TallySkipSpans appends to the predictor's list for *every* service it solves, skip budget
    or not (traceweaver_v3.py:1141, :976-987): [(start, end, 30)]."""
    n = unit.n_in
    cut = np.arange(BATCH_MIS, n - 1, BATCH_MIS)                          # i % 30 == 0, i != 0, i != n - 1
    edges = np.concatenate([[unit.in_start[0]], unit.in_end[cut], [np.max(unit.in_end)]]).astype(np.int64)
    return [(int(a), int(b), BATCH_MIS) for a, b in zip(edges[:-1], edges[1:])]


def tally_skip_spans(unit, prior_windows=()):
    """TallySkipSpans (traceweaver_v3.py:853-989).  prior_windows: the windows of services the same predictor object
    solved before (the reference never clears self.time_windows, SURVEY.md hazard H8)."""
    n, E = unit.n_in, unit.E
    tw = list(prior_windows) + time_windows(unit)
    keys = sorted(tw, key=lambda w: w[0])
    budget = np.array([n - int(unit.out_off[e + 1] - unit.out_off[e]) for e in range(E)], dtype=np.int64)
    pool = np.zeros((E, len(keys)), dtype=np.int32)
    expected = np.array([k[2] for k in keys], dtype=np.float64)
    for e in range(E):
        if budget[e] <= 0:
            continue
        st = np.sort(unit.out_start[unit.out_off[e]:unit.out_off[e + 1]])
        existing = np.array([np.searchsorted(st, b, side="right") - np.searchsorted(st, a, side="right") for a, b, _ in keys], dtype=np.float64)
        order = np.argsort(existing)[::-1]
        srt = existing[order]
        csum = np.cumsum(srt)
        lam, rest = 0.0, 0.0
        for i in range(len(keys)):                                        # the water level
            lam = (budget[e] + csum[i]) // (i + 1)
            rest = (budget[e] + csum[i]) % (i + 1)
            if lam <= srt[i]:
                break
        want = np.maximum(lam - srt, 0)
        give = np.minimum(want, expected - srt)        # positional: expected[i] next to the i-th largest, as the reference does
        alloc = np.zeros(len(keys))
        alloc[order] = give
        rest += float(np.sum(want - give))
        while rest > 0:
            changed = False
            for i in reversed(range(len(keys))):
                if rest > 0 and alloc[order[i]] < expected[i] - srt[i]:
                    alloc[order[i]] += 1
                    rest -= 1
                    changed = True
            if not changed:
                break
        pool[e] = np.maximum(alloc, 0).astype(np.int32)
    return keys, budget, pool


def build_distributions(engine, unit):
    """BuildDistributions (traceweaver_v3.py:108-172): sweep on the device, (mean, std) per pair on the host."""
    E, n = unit.E, unit.n_in
    start = np.concatenate([unit.in_start, unit.out_start])
    dur = np.concatenate([unit.in_end - unit.in_start, unit.out_end - unit.out_start])
    ep = np.concatenate([np.zeros(n, np.uint8)] + [np.full(int(unit.out_off[e + 1] - unit.out_off[e]), 1 + e, np.uint8) for e in range(E)])
    order = np.argsort(start, kind="stable")
    large = int(np.max(unit.in_end - unit.in_start))
    key, val = engine.build_distributions(start[order], dur[order], ep[order], large, E)
    tab = np.full((E + 1, E + 1, 2), np.nan)
    for k in np.unique(key[key >= 0]):
        v = val[key == k]
        tab[k // (E + 1), k % (E + 1)] = (np.mean(v), np.std(v))
    d = dur[order][ep[order] == 0]
    tab[0, 0] = (np.mean(d), np.std(d))                                   # the requests' own durations (never read by the scorer)
    return tab, large


def plan(engine, unit, prior_windows=()):
    keys, budget, pool = tally_skip_spans(unit, prior_windows)
    dist, large = build_distributions(engine, unit)
    return SkipPlan(keys, budget, pool, dist, large)


def decode(idx):
    """Raw candidate indices -> (index with skip spans as -2 - position in the pool, time window or -1)."""
    from . import _ffi

    idx = np.asarray(idx)
    skip = idx <= -_ffi.TW_SKIP_BASE
    code = np.where(skip, -idx - _ffi.TW_SKIP_BASE, 0)
    return np.where(skip, -2 - code % _ffi.TW_SKIP_STRIDE, idx), np.where(skip, code // _ffi.TW_SKIP_STRIDE, -1)
