import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

# ref_*: reference runs on the shipped corpora; refsyn_*: reference runs on synthetic heavy-load units
# (size-capped windows, span consumption across windows) -- oracle/refrun/gen_golden*.py
GOLDEN = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "ref_*.npz"))) + \
    sorted(glob.glob(os.path.join(REPO, "tests", "golden", "refsyn_*.npz")))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_ids():
    return [os.path.basename(f)[:-4].split("_", 1)[1] for f in GOLDEN]


@pytest.fixture(scope="session")
def oracle():
    import tw_oracle

    tw_oracle.lib()
    return tw_oracle


@pytest.fixture(scope="session")
def emu_lib():
    """Engine source compiled for the host against tests/hostemu (logic checks without a GPU)."""
    os.environ["TW_TILE"] = "1"
    os.environ["TW_COOP_THREADS"] = "1"
    os.environ.setdefault("TW_STAGE_MIN_TILES", "0")   # the small units of the CPU tier take the per-class window / selection stages too
    from tests.hostemu.build_emu import build

    return build()


def unit_from_golden(d):
    import tw_oracle as T
    from traceweaver_amd.engine import UnitArrays

    svc = T.service_from_golden(d)
    return svc, UnitArrays(svc.in_start, svc.in_end, svc.out_off, svc.out_start, svc.out_end, svc.dag, svc.key_rank)


def golden_mixtures(d):
    return np.maximum(d["mix_n"], 0).astype(np.int32), np.ascontiguousarray(d["mix_p"][:, :, [0, 1, 3]])


def assert_pass_equal(res, ora, end_flag, tag=""):
    """Engine result dict vs oracle pass dict: indices bit-exact, scores bit-exact."""
    assert np.array_equal(res["window_end"], end_flag), tag + " window ends"
    assert np.array_equal(res["topk_n"], ora["topk2_n"]), tag + " candidate counts"
    assert np.array_equal(np.transpose(res["topk_idx"], (2, 0, 1)), ora["topk2_idx"]), tag + " top-5 tuples"
    m = ~np.isnan(ora["topk2_score"])
    assert np.array_equal(res["topk_score"].T[m], ora["topk2_score"][m]), tag + " top-5 scores"
    assert np.array_equal(res["chosen"], ora["chosen"]), tag + " selection"
    assert np.array_equal(res["parent"], ora["parent"]), tag + " parent arrays"
    assert np.array_equal(res["leaves"], ora["leaves"]), tag + " enumerated tuples"
    assert res["not_best_count"] == ora["not_best_count"], tag + " not_best_count"
    assert res["cnt_unassigned"] == ora["cnt_unassigned"], tag + " cnt_unassigned"
    assert res["n_windows"] == ora["n_windows"], tag + " window count"


def skip_tie_requests(name):
    """Requests of a frozen skip-mode run (refskip_<...>__frontend) that lie in a window where the exact selection is proven a
    near-tie of the frozen run's (tests/golden/skip_tie_windows.json, written by tests/golden/make_skip_tie_windows.py): only
    there may the engine's assignment differ from the frozen run's."""
    import json

    with open(os.path.join(REPO, "tests", "golden", "skip_tie_windows.json")) as f:
        return set(json.load(f)[name])
