"""Skip mode through the C-ABI (tw_batch.skip): the engine's one pass with skip spans against the oracle -- bit for bit:
windows, both top-5 lists incl. which skip span of which time window, scores, tuple counts, selections, the final
assignment with -2 = ('Skip','Skip') -- on the inputs of the reference runs with --cache_rate 0.05 ... 0.3 (which pin the
oracle, tests/test_skip_oracle.py); the host side (cache-hit transform, time windows / water-filling, BuildDistributions
with its sweep on the device) against what the reference computed in those runs."""
import glob
import os

import numpy as np
import pytest

from conftest import REPO

SKIP_GOLDEN = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "refskip_*__frontend.npz")))


def unit_of(d, prefix=""):
    import tw_oracle as T
    from traceweaver_amd.engine import UnitArrays

    st, du = d[prefix + "out_start"], d[prefix + "out_dur"]
    return UnitArrays(d[prefix + "in_start"], d[prefix + "in_start"] + d[prefix + "in_dur"], d[prefix + "out_off"], st, st + du, d["dag"],
                      T.golden_key_rank(d))


def check(lib_path, path):
    import tw_oracle as T
    from traceweaver_amd import skipmode
    from traceweaver_amd.engine import Engine

    d = np.load(path)
    eng = Engine(0, lib_path=lib_path)
    # the executor's transform: create_cache_hits on the inputs of the untransformed run
    pre = unit_of(d, "pre_")
    unit, truth, kept = skipmode.cache_hits(pre, d["pre_true_parent"], float(d["cache_rate"]))
    assert int(kept.sum()) == unit.out_off[1]
    for name in ("in_start", "in_end", "out_off", "out_start", "out_end"):
        want = {"in_end": d["in_start"] + d["in_dur"], "out_end": d["out_start"] + d["out_dur"]}.get(name, d[name] if name in d.files else None)
        assert np.array_equal(getattr(unit, name), want), name
    assert np.array_equal(truth, d["true_parent"])
    # TallySkipSpans / WaterFill and BuildDistributions
    sp = skipmode.plan(eng, unit)
    assert np.array_equal(np.array(sp.windows), d["time_windows"]) and np.array_equal(sp.budget, d["skip_budget"])
    assert np.array_equal(sp.pool, d["skip_count"]) and sp.large_delay == int(d["large_delay"])
    m = ~np.isnan(d["dist_table"])
    assert np.array_equal(np.isnan(sp.dist), ~m) and np.array_equal(sp.dist[m], d["dist_table"][m])
    # the pass
    eng.load([unit], skip=[sp])
    eng.set_truth([truth])
    eng.run_pass1()
    r = eng.results(1)[0]
    ev = eng.evaluate()[0]
    with pytest.raises(Exception):
        eng.fit_mixtures()                       # a skip-mode batch runs one pass
    eng.close()
    svc = T.service_from_golden(d)
    end_flag, pre_leaves, win = T.windows(svc)
    o = T.run_skip(svc, end_flag, sp.windows, sp.pool, sp.dist)
    assert np.array_equal(r["window_end"], end_flag)
    assert np.array_equal(r["topk_n"], o["topk2_n"])
    assert np.array_equal(np.transpose(r["topk_idx"], (2, 0, 1)), o["topk2_idx"])
    ok = ~np.isnan(o["topk2_score"])
    assert np.array_equal(r["topk_score"].T[ok], o["topk2_score"][ok])
    assert np.array_equal(r["chosen"], o["chosen"]) and np.array_equal(r["parent"], o["parent"])
    assert np.array_equal(r["leaves"], o["leaves"])
    assert (r["not_best_count"], r["cnt_unassigned"], r["n_windows"]) == (o["not_best_count"], o["cnt_unassigned"], o["n_windows"])
    assert r["budget_windows"] == 0
    # against the reference run itself: the lists (skip spans decoded), and the assignment up to the solver's tolerance
    idx, w = skipmode.decode(np.transpose(r["topk_idx"], (2, 0, 1)))
    assert np.array_equal(idx, d["p0_topk2_idx"]) and np.array_equal(w, d["p0_topk2_win"])
    from conftest import skip_tie_requests

    differing = set(np.flatnonzero((r["parent"] != d["final_parent"]).any(axis=0)).tolist())
    assert differing <= skip_tie_requests(os.path.basename(path)[:-4])   # only inside the proven near-tie windows
    assert (r["parent"] == -2).sum() > 0
    assert ev["correct"] == int(np.all(r["parent"] == truth, axis=0).sum())
    assert abs(ev["accuracy"] - float(np.all(d["final_parent"] == truth, axis=0).mean())) <= 0.005


@pytest.mark.parametrize("path", SKIP_GOLDEN, ids=[os.path.basename(p)[8:-4] for p in SKIP_GOLDEN])
def test_skip_mode_emulated(emu_lib, oracle, path):
    check(emu_lib, path)


@pytest.mark.gpu
@pytest.mark.parametrize("path", SKIP_GOLDEN, ids=[os.path.basename(p)[8:-4] for p in SKIP_GOLDEN])
def test_skip_mode_gpu(oracle, path):
    check(None, path)


def test_no_skip_batches_still_reject_short_endpoints(emu_lib):
    from traceweaver_amd.engine import Engine, EngineError

    d = np.load(SKIP_GOLDEN[0])
    eng = Engine(0, lib_path=emu_lib)
    with pytest.raises(EngineError) as ex:
        eng.load([unit_of(d)])
    assert ex.value.code == -2
    eng.close()
