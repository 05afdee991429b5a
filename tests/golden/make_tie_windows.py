"""Writes tests/golden/tie_windows.json: per frozen reference run (ref_*.npz) and pass, the incoming spans that lie in a
window whose optimum is not unique -- the oracle's exact selection differs there from the frozen run's (HiGHS standing in
for Gurobi) and both selections are proven to be independent sets of equal weight (tests/test_oracle_golden.py::_tie_windows,
run with the frozen selections committed so that a tie does not cascade).  The GPU tier then accepts a difference between
the engine's parent arrays and the frozen run's only inside these windows (tests/test_gpu_parity.py).

    python tests/golden/make_tie_windows.py        (needs only the oracle, not the reference)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)


def main():
    import tw_oracle as oracle
    from conftest import GOLDEN, golden_mixtures
    from test_oracle_golden import _tie_windows

    out = {}
    for path in GOLDEN:
        d = np.load(path)
        if str(d["dataset"]).startswith("synthetic"):
            continue
        svc = oracle.service_from_golden(d)
        end_flag, _, win = oracle.windows(svc)
        p1 = oracle.run_pass(svc, end_flag, gauss=oracle.gauss_params(svc), forced=d["p0_chosen"])
        mix_n, mix_p = golden_mixtures(d)
        p2 = oracle.run_pass(svc, end_flag, mix_n=mix_n, mix_p=mix_p, forced=d["p1_chosen"])
        t1, t2 = sorted(int(i) for i in _tie_windows(d, 0, p1, win)), sorted(int(i) for i in _tie_windows(d, 1, p2, win))
        if t1 or t2:
            out[os.path.basename(path)[:-4]] = {"pass1": t1, "pass2": t2}
    with open(os.path.join(HERE, "tie_windows.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("%d of %d frozen runs hold a tied window; %d + %d spans" % (
        len(out), len(GOLDEN), sum(len(v["pass1"]) for v in out.values()), sum(len(v["pass2"]) for v in out.values())))


if __name__ == "__main__":
    main()
