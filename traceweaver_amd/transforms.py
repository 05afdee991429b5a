"""Load scaling on index arrays: the reference's `repeat_change_spans` (helpers/transforms.py:10-40), the transform
`--compress_factor N` applies to every service before the predictor runs (executor.py:1086-1097,1146-1148; the route
exps/exp5 takes to raise the load of the Alibaba call graphs).

What the reference does per service, with load factor f = max(1, ceil(compress_factor / #replicas)):
  partitions sorted by trace id; per request   x = in.start / f                 (Python true division: a float)
                                               out.start = x + (out.start - in.start)   for every endpoint
                                               in.start = x;   durations untouched
  partitions re-sorted by (start, start + duration)  (stable: equal keys stay in trace-id order);
  ground truth rebuilt (helpers/utils.py:22-32).  The call-order DAG was inferred before (executor.py:1143).

From there on every timestamp the predictor sees is a binary64 float.  The engine works on int64 timestamps, so a
scaled unit is handed over *exactly*: all its float timestamps (starts and the ends fl(start + duration) the reference
forms) are integer multiples of 2^-k for some small k; they are passed as those integers with time_scale = 2^-k
(UnitArrays.time_scale -> tw_batch.unit_time_scale).  Comparisons, sort order and ties are then the reference's float
comparisons, differences are exact, and the engine sums timestamps in binary64 as Python's sum() does over floats --
pass 1 reproduces the reference bit for bit on such inputs (tests/test_transforms.py against runs frozen from the
reference, tests/golden/refcmp_*.npz).

One deliberate difference after pass 1: the reference's refit (traceweaver_v3.py:717-762) looks the chosen outgoing span
up in `self.all_spans` -- the *untransformed* originals -- and subtracts the transformed incoming timestamp, so its
pass-2 mixtures are fitted to differences of the order of the epoch and its final accuracy collapses (0 % on
hotel_load50 x2 where pass 1 reaches 99 %; on other corpora every fit raises and the run stops).  The engine's gap
samples use the timestamps it was given.  Given the reference's own (frozen) mixture tables pass 2 is reproduced too.
"""
import math

import numpy as np

from .engine import UnitArrays


def load_factor(compress_factor, n_replicas):
    """executor.py:1089-1091."""
    return max(1, math.ceil(compress_factor / max(int(n_replicas), 1)))


def exact_binary_exponent(arrays):
    """Smallest k >= 0 such that every value of the float64 arrays is an integer multiple of 2^-k."""
    k = 0
    for a in arrays:
        a = np.asarray(a, dtype=np.float64)
        a = a[a != 0]
        if len(a) == 0:
            continue
        m, e = np.frexp(a)                                     # a = m * 2^e, 0.5 <= |m| < 1
        mant = np.abs(np.ldexp(m, 53)).astype(np.int64)        # the 53-bit integer mantissa
        tz = np.log2((mant & -mant).astype(np.float64)).astype(np.int64)   # its trailing zero bits
        k = max(k, int(np.max(53 - e - tz)))
    return k


def to_exact_units(float_arrays):
    """float64 arrays -> (int64 arrays, time_scale) with value == int * time_scale exactly."""
    k = exact_binary_exponent(float_arrays)
    top = max((float(np.max(np.abs(a))) for a in float_arrays if len(a)), default=0.0)
    if top * 2.0 ** k >= 2.0 ** 62:
        raise ValueError("timestamps span too many binades to be represented exactly in int64 (2^-%d units up to %g)" % (k, top))
    out = [np.ldexp(np.asarray(a, dtype=np.float64), k).astype(np.int64) for a in float_arrays]
    for a, b in zip(float_arrays, out):
        if not np.array_equal(np.ldexp(b.astype(np.float64), -k), a):
            raise AssertionError("inexact timestamp conversion")
    return out, float(np.ldexp(1.0, -k))


class ScaledUnit(object):
    """Result of compress_unit: `arrays` (UnitArrays with time_scale) in the new (start, end) order, `true_parent`
    [E, n_in] in that order, `in_perm` / `out_perm[e]` = old index of every new position, and the float64 timestamps
    (`in_start`, `out_start`; durations are unchanged) as the reference's predictor would see them."""

    def __init__(self, arrays, true_parent, in_perm, out_perm, in_start, out_start, factor):
        self.arrays, self.true_parent, self.in_perm, self.out_perm = arrays, true_parent, in_perm, out_perm
        self.in_start, self.out_start, self.factor = in_start, out_start, factor


def compress_unit(arrays, true_parent, factor, trace_key=None):
    """repeat_change_spans (helpers/transforms.py:10-40) for one unit in integer microseconds.

    arrays      UnitArrays, int64 microseconds, every list sorted by (start, end)
    true_parent [E, n_in] index of the request's own call at every endpoint (the reference pairs position i of every
                partition after sorting by trace id and asserts equal trace ids, helpers/transforms.py:25-29)
    factor      the service's load factor (an int >= 1, see load_factor())
    trace_key   [n_in] sortable keys in the order of the trace ids (the ids themselves, or their ranks); only decides
                the order of spans whose transformed (start, end) are equal.  None = current order.
    """
    if arrays.time_scale is not None:
        raise ValueError("unit is already load-scaled")
    tp = np.asarray(true_parent)
    n, E = arrays.n_in, arrays.E
    if tp.shape != (E, n) or (tp < 0).any():
        raise ValueError("load scaling needs the request's own call at every endpoint (helpers/transforms.py:25-29 asserts it)")
    factor = int(factor)
    by_trace = np.arange(n) if trace_key is None else np.argsort(np.asarray(trace_key), kind="stable")
    x = arrays.in_start.astype(np.float64) / float(factor)                      # transforms.py:21
    in_dur = (arrays.in_end - arrays.in_start).astype(np.float64)
    in_end = x + in_dur                                                         # fl(start + duration), as every use forms it

    def order(start, end, first):
        o = first[np.argsort(end[first], kind="stable")]                        # list.sort(key=(start, end)) is stable
        return o[np.argsort(start[o], kind="stable")]

    in_perm = order(x, in_end, by_trace)
    new_pos = np.empty(n, dtype=np.int64)
    new_pos[in_perm] = np.arange(n)
    out_start, out_end, out_perm, truth = [], [], [], np.empty((E, n), dtype=np.int32)
    for e in range(E):
        a, b = int(arrays.out_off[e]), int(arrays.out_off[e + 1])
        if b - a != n or len(np.unique(tp[e])) != n:
            raise ValueError("load scaling needs exactly one call per request at every endpoint")
        os_, oe_ = arrays.out_start[a:b], arrays.out_end[a:b]
        own = tp[e]                                                             # request i -> its call at endpoint e
        y = np.empty(n, dtype=np.float64)
        y[own] = x + (os_[own] - arrays.in_start).astype(np.float64)            # transforms.py:30
        ye = y + (oe_ - os_).astype(np.float64)
        first = own[by_trace]                                                   # the partition in trace-id order
        perm = order(y, ye, first)
        pos = np.empty(n, dtype=np.int64)
        pos[perm] = np.arange(n)
        truth[e, new_pos] = pos[own]                                            # GetGroundTruth after the re-sort
        out_start.append(y[perm]); out_end.append(ye[perm]); out_perm.append(perm)
    fl = [x[in_perm], in_end[in_perm], np.concatenate(out_start), np.concatenate(out_end)]
    (i_s, i_e, o_s, o_e), scale = to_exact_units(fl)
    scaled = UnitArrays(i_s, i_e, arrays.out_off, o_s, o_e, arrays.dag, arrays.key_rank, time_scale=scale)
    scaled.in_dur = (arrays.in_end - arrays.in_start)[in_perm]                  # the integer durations (the baselines read them)
    return ScaledUnit(scaled, truth, in_perm, out_perm, fl[0], fl[2], factor)
