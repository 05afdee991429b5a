"""The oracle's strict-IEEE log/exp/log1p (fdlibm algorithms) against the host libm, and its numpy
summation / variance emulation against numpy itself."""
import ctypes
import math

import numpy as np


def _ulps(a, b):
    return abs(int(np.float64(a).view(np.int64)) - int(np.float64(b).view(np.int64)))


def test_elementary_functions_within_one_ulp(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    worst = [0, 0, 0]
    for _ in range(20000):
        x = float(10.0 ** rng.uniform(-300, 300))
        worst[0] = max(worst[0], _ulps(L.two_log(x), math.log(x)))
        y = float(rng.uniform(-740, 700))
        worst[1] = max(worst[1], _ulps(L.two_exp(y), math.exp(y)))
        z = float(rng.uniform(0, 8)) if rng.random() < 0.5 else float(10.0 ** rng.uniform(-30, 1))
        worst[2] = max(worst[2], _ulps(L.two_log1p(z), math.log1p(z)))
    assert worst[0] <= 1 and worst[1] <= 1 and worst[2] <= 1, worst


def test_elementary_special_values(oracle):
    L = oracle.lib()
    assert L.two_log(1.0) == 0.0 and L.two_log1p(0.0) == 0.0 and L.two_exp(0.0) == 1.0
    assert L.two_exp(-math.inf) == 0.0 and math.isinf(L.two_log(0.0)) and math.isnan(L.two_log(-1.0))
    assert L.two_log(0.001) == math.log(0.001) or _ulps(L.two_log(0.001), math.log(0.001)) <= 1


def test_gauss_params_match_numpy_statistics(oracle):
    """ComputeDistParams (traceweaver_v3.py:590-617) restated with numpy on random rank-aligned arrays:
    mean exact, std = sqrt(batch) * sqrt(np.var(batch_means, ddof=1)) bit-exact (scipy 1.14 tstd)."""
    rng = np.random.default_rng(1)
    for n in (2, 7, 10, 25, 99, 100, 230):
        in_start = np.sort(rng.integers(0, 10**7, n)) + 10**15
        dur = rng.integers(1000, 50000, n)
        o_start = np.sort(in_start + rng.integers(1, 500, n))
        o_dur = rng.integers(1, 900, n)
        svc = oracle.Service(in_start, dur, [0, n], o_start, o_dur, [[0]])
        g = oracle.gauss_params(svc)
        t_in_end = np.sort(in_start + dur)
        t_out_end = np.sort(o_start + o_dur)
        for b in range(g.shape[0]):
            a, z = b * 100, min(n, b * 100 + 100)
            for slot, (t1, t2) in ((0, (in_start, o_start)), (2, (t_out_end, t_in_end))):
                x1, x2 = [int(v) for v in t1[a:z]], [int(v) for v in t2[a:z]]
                mean = (sum(x2) - sum(x1)) / len(x1)
                bs = math.ceil(len(x1) / 10)
                bm = [(sum(x2[i:i + bs]) - sum(x1[i:i + bs])) / len(x1[i:i + bs]) for i in range(0, len(x1), bs)]
                assert g[b, slot, 0] == mean
                if len(bm) > 1:
                    std = math.sqrt(bs) * float(np.sqrt(np.var(np.array(bm), ddof=1)))
                    assert g[b, slot, 1] == std
                else:
                    assert math.isnan(g[b, slot, 1])
