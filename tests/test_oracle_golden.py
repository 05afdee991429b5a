"""Pins the CPU oracle to golden vectors frozen from the *reference* (oracle/refrun/gen_golden.py ran
the unmodified TraceWeaverV3 on the shipped Jaeger corpora).  Index outputs must be identical; float
scores agree to ~1e-15 relative (the oracle uses fdlibm-style log/exp, numpy/scipy use their own libm)."""
import numpy as np
import pytest

from conftest import GOLDEN, golden_ids, golden_mixtures

SCORE_RTOL = 1e-12


@pytest.fixture(scope="module", params=GOLDEN, ids=golden_ids())
def case(request, oracle):
    """Both passes are run with the frozen run's selections committed ("teacher forcing"): where an optimum
    is not unique (exact ties, the closed-source Gurobi / HiGHS choice is unspecified) a different but
    equally good pick would otherwise change which spans later windows may use and the comparison would
    cascade.  The oracle's *own* selection per window is still reported and checked (_tie_windows)."""
    d = np.load(request.param)
    svc = oracle.service_from_golden(d)
    end_flag, pre, win = oracle.windows(svc)
    g = oracle.gauss_params(svc)
    p1 = oracle.run_pass(svc, end_flag, gauss=g, forced=d["p0_chosen"])
    mix_n, mix_p = golden_mixtures(d)
    p2 = oracle.run_pass(svc, end_flag, mix_n=mix_n, mix_p=mix_p, forced=d["p1_chosen"])
    return d, svc, end_flag, pre, win, g, p1, p2


def test_windows(case):
    d, svc, end_flag, pre, win, g, p1, p2 = case
    assert np.array_equal(win, d["windows"])            # CreateWindows2, traceweaver_v3.py:1020-1078
    assert np.array_equal(pre, d["pre_leaves"])         # DfsTraverse3 tuple counts
    assert int(end_flag.sum()) == len(set(d["windows"][:, 1].tolist()))


def test_gauss_params(case):
    d, svc, end_flag, pre, win, g, p1, p2 = case
    scored = ~np.isnan(g[..., 0])
    assert scored.any()
    ref = d["params3"]
    assert not np.isnan(ref[scored]).any()
    assert np.allclose(g[scored], ref[scored], rtol=1e-14, atol=0)


@pytest.mark.parametrize("p", [0, 1])
def test_topk(case, p):
    d, svc, end_flag, pre, win, g, p1, p2 = case
    o = (p1, p2)[p]
    for kind in ("topk", "topk2"):
        assert np.array_equal(o[kind + "_n"], d["p%d_%s_n" % (p, kind)])
        assert np.array_equal(o[kind + "_idx"], d["p%d_%s_idx" % (p, kind)])
        ref = d["p%d_%s_score" % (p, kind)]
        m = ~np.isnan(ref)
        assert np.array_equal(m, ~np.isnan(o[kind + "_score"]))
        assert np.allclose(o[kind + "_score"][m], ref[m], rtol=SCORE_RTOL, atol=0)


def _tie_windows(d, p, o, windows):
    """Windows where the oracle's exact selection differs from the frozen run.  The goldens were made with
    HiGHS standing in for Gurobi; when the optimum is not unique any exact solver may return either
    optimum (SURVEY.md 8(c) "parity unpinned").  A difference is accepted only if both selections are
    independent sets of the same conflict graph with equal total weight."""
    ref_ch, idx, sc = d["p%d_chosen" % p], d["p%d_topk_idx" % p], d["p%d_topk_score" % p]
    bad = np.nonzero(o["chosen"] != ref_ch)[0]
    tie_spans = set()
    for i in bad:
        if i in tie_spans:
            continue
        a, b = [(int(s0), int(s1)) for s0, s1 in windows if s0 <= i <= s1][0]
        tot = []
        for ch in (ref_ch, o["chosen"]):
            used, w = set(), 0.0
            for j in range(a, b + 1):
                if ch[j] < 0:
                    continue
                for e, x in enumerate(idx[j, ch[j]]):
                    assert (e, int(x)) not in used, "selection is not an independent set"
                    used.add((e, int(x)))
                w += 10000.0 + sc[j, ch[j]]
            tot.append(w)
        assert abs(tot[0] - tot[1]) <= 1e-9 * abs(tot[0]), "window %d-%d: selections differ and are not tied: %r" % (a, b, tot)
        tie_spans.update(range(a, b + 1))
    return tie_spans


def test_selection_and_assignment(case):
    d, svc, end_flag, pre, win, g, p1, p2 = case
    n = svc.n_in
    ties1 = _tie_windows(d, 0, p1, win)
    ties2 = _tie_windows(d, 1, p2, win)
    heavy = str(d["dataset"]).startswith("synthetic")
    # exact ties are rare on the microsecond corpora; the ms-granular heavy-load units are saturated with them
    assert heavy or (len(ties1) <= 0.01 * n and len(ties2) <= 0.01 * n)
    keep1 = np.array([i not in ties1 for i in range(n)])
    keep2 = np.array([i not in ties2 for i in range(n)])
    assert np.array_equal(p1["chosen"][keep1], d["p0_chosen"][keep1])
    assert np.array_equal(p2["chosen"][keep2], d["p1_chosen"][keep2])
    # the committed (frozen) selections reproduce the frozen parent arrays and counters exactly
    assert np.array_equal(p1["parent"], d["pass1_parent"])
    assert np.array_equal(p2["parent"], d["final_parent"])
    assert p2["not_best_count"] == int(d["not_best_count"])
    assert p2["cnt_unassigned"] == int(d["cnt_unassigned"])
    assert np.array_equal(p1["leaves"] + p2["leaves"], d["per_span_candidates"])
    assert np.array_equal(np.transpose(d["final_topk"], (1, 2, 0)), p2["topk2_idx"])


def test_gap_samples_and_final_refit(case, oracle):
    """traceweaver_v3.py:717-786: gaps of the pass-1 assignment, and the deterministic final refit
    (random_state=100) given the component count the reference selected."""
    d, svc, end_flag, pre, win, g, p1, p2 = case
    gaps = oracle.gaps(svc, d["pass1_parent"])
    checked = 0
    for q, samples in enumerate(gaps):
        if samples is None or d["mix_n"][q] <= 0:
            continue
        n, p = oracle.fit_mixture(samples, n_selected=int(d["mix_n"][q]))
        ref = d["mix_p"][q, :n]
        assert np.allclose(p[:n, 0], ref[:, 0], rtol=1e-9) and np.allclose(p[:n, 1], ref[:, 1], rtol=1e-9)
        assert np.allclose(p[:n, 2], ref[:, 3], rtol=1e-9)
        checked += 1
        if checked >= 3:
            break
    assert checked > 0


def test_tie_window_list_is_current(case):
    """tests/golden/tie_windows.json (what the GPU tier accepts as differences from the frozen runs) holds exactly the tied
    windows this module proves."""
    import json
    import os

    d, svc, end_flag, pre, win, g, p1, p2 = case
    if str(d["dataset"]).startswith("synthetic"):
        pytest.skip("tie-saturated synthetic unit: compared with the oracle only")
    name = "ref_%s__%s" % (str(d["dataset"]), str(d["process"]))
    with open(os.path.join(os.path.dirname(GOLDEN[0]), "tie_windows.json")) as f:
        listed = json.load(f)
    t = listed.get(name, {"pass1": [], "pass2": []})
    assert sorted(int(i) for i in _tie_windows(d, 0, p1, win)) == t["pass1"]
    assert sorted(int(i) for i in _tie_windows(d, 1, p2, win)) == t["pass2"]
