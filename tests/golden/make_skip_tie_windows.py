"""Writes tests/golden/skip_tie_windows.json: per frozen skip-mode run (refskip_*__frontend.npz) the requests that lie in a
window where the canonical exact selection (oracle) differs from the frozen run's, each proven a near-tie: both selections
assign the same number of requests and their weights differ by less than the MILP solver's optimality tolerance (the scores
are densities of the order of 1e-4 next to the 10000 offset of a node weight, traceweaver_v3.py:1260; HiGHS standing in for
Gurobi returns selections up to 4e-8 below the optimum, the oracle takes the exact optimum on integer weights).  The
candidate lists of every request -- both top-5 lists incl. which skip span they hold -- are identical to the frozen run's
(tests/test_skip_oracle.py), so a near-tie does not cascade into later windows.  The engine equals the oracle bit for bit
(tests/test_skip_engine.py); tests then accept a difference between the engine and the frozen run only inside these windows.

    python tests/golden/make_skip_tie_windows.py        (needs only the oracle, not the reference)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

TOL = 1.0e-6   # weights of the two selections of a listed window (sums of 10000 + score over the assigned requests)


def near_tie_requests(oracle, d):
    from test_skip_oracle import solve

    svc, keys, budget, pool, tab, large, end_flag, pre, win, o = solve(oracle, d)
    out, first = [], 0
    for i in range(svc.n_in):
        if not end_flag[i]:
            continue
        a, b = o["chosen"][first:i + 1], d["p0_chosen"][first:i + 1]
        if not np.array_equal(a, b):
            wa = sum(10000.0 + o["topk_score"][j, o["chosen"][j]] for j in range(first, i + 1) if o["chosen"][j] >= 0)
            wb = sum(10000.0 + d["p0_topk_score"][j, d["p0_chosen"][j]] for j in range(first, i + 1) if d["p0_chosen"][j] >= 0)
            assert int((a >= 0).sum()) == int((b >= 0).sum()), "window %d..%d: another number of requests assigned" % (first, i)
            assert abs(wa - wb) < TOL and wa >= wb - 1e-9, "window %d..%d: not a near-tie (%r vs %r)" % (first, i, wa, wb)
            out += list(range(first, i + 1))
        first = i + 1
    return out


def main():
    import glob

    import tw_oracle as oracle

    res = {}
    for path in sorted(glob.glob(os.path.join(HERE, "refskip_*__frontend.npz"))):
        res[os.path.basename(path)[:-4]] = near_tie_requests(oracle, np.load(path))
    with open(os.path.join(HERE, "skip_tie_windows.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in res.items()})


if __name__ == "__main__":
    main()
