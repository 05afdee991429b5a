"""The two neighbours of the hot path that work on the resident arrays: FindOrder
(executor.py:214-285) and the accuracy reductions (helpers/utils.py:62-145), device vs numpy restatement."""
import numpy as np
import pytest

import parity
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine

CASES = [(41, 900, "chain3", 2, 1), (42, 700, "par4", 1.5, 1), (43, 500, "diamond", 3, 1000), (44, 300, "mix8", 1.3, 1),
         (45, 1200, "single", 4, 1), (46, 650, "fan6", 1.5, 1)]


def topk_ok(res, tp):
    idx, n = res["topk_idx"], res["topk_n"]            # [5, E, n], [n]
    same = np.all(idx == tp[None, :, :], axis=1)       # [5, n]
    valid = np.arange(idx.shape[0])[:, None] < n[None, :]
    return np.any(same & valid, axis=0)


def check(lib_path):
    units, truth = parity.stress_units(CASES)
    eng = Engine(0, lib_path=lib_path)
    # FindOrder: the generator derives every unit's DAG from the true assignments the same way
    dags = eng.find_order(units, truth)
    for u, d in zip(units, dags):
        assert np.array_equal(d, u.dag)
    # ... also with endpoints in a non-topological order (as the executor calls it, partition-key order)
    u, tp = units[0], truth[0]
    perm = [2, 0, 1]
    n = u.n_in
    from traceweaver_amd.engine import UnitArrays
    shuffled = UnitArrays(u.in_start, u.in_end, np.arange(4) * n,
                          np.concatenate([u.out_start[u.out_off[e]:u.out_off[e + 1]] for e in perm]),
                          np.concatenate([u.out_end[u.out_off[e]:u.out_off[e + 1]] for e in perm]), np.zeros((3, 3)))
    d = eng.find_order([shuffled], [tp[perm]])[0]
    assert np.array_equal(d, u.dag[np.ix_(perm, perm)])
    # accuracy reductions after pass 1
    rng = np.random.default_rng(5)
    n_traces = 400
    traces = [rng.integers(0, n_traces, u.n_in).astype(np.int32) for u in units]
    traces[1][:10] = -1                                # spans outside every trace are ignored
    eng.load(units)
    eng.set_truth(truth, traces, n_traces)
    eng.run_pass1()
    res = eng.results(1)
    per, e2e, flags = eng.evaluate(trace_flags=True)
    bad = np.zeros(n_traces, bool)
    bad_k = np.zeros(n_traces, bool)
    for u, tp, r, ev, tr in zip(units, truth, res, per, traces):
        ok = np.all(r["parent"] == tp, axis=0)
        okk = topk_ok(r, tp)
        assert ev["n_in"] == u.n_in and ev["correct"] == int(ok.sum()) and ev["correct_topk"] == int(okk.sum())
        assert ev["unassigned"] == int((r["parent"] < 0).any(axis=0).sum()) == r["cnt_unassigned"]
        assert ev["accuracy"] == pytest.approx(synth.accuracy(r["parent"], tp))
        m = tr >= 0
        bad[tr[m & ~ok]] = True
        bad_k[tr[m & ~okk]] = True
    assert np.array_equal(flags[0].astype(bool), bad) and np.array_equal(flags[1].astype(bool), bad_k)
    assert e2e == (int((~bad).sum()), int((~bad_k).sum()))
    eng.close()


def check_find_order_on_reference_runs(lib_path):
    """FindOrder on the device against the call-order DAGs the reference's own FindOrder (executor.py:214-285) produced
    in the frozen runs -- all 90 services of the shipped corpora plus the synthetic reference runs -- from the inputs and
    ground truth of those runs; also with the endpoints in partition-key order, as the executor calls it."""
    from conftest import GOLDEN, unit_from_golden
    from traceweaver_amd.engine import UnitArrays

    ds = [np.load(p) for p in GOLDEN]
    units = [unit_from_golden(d)[1] for d in ds]
    eng = Engine(0, lib_path=lib_path)
    dags = eng.find_order(units, [d["true_parent"] for d in ds])
    checked = 0
    for d, u, got in zip(ds, units, dags):
        assert np.array_equal(got, d["dag"]), "%s / %s" % (d["dataset"], d["process"])
        checked += int(d["dag"].sum() > 0)
    assert checked >= 10                       # chains, diamonds and the transitive edges of the hotel frontend are among them
    shuf, truth, want = [], [], []
    for d, u in zip(ds, units):
        if u.E < 2:
            continue
        keys = [str(x) for x in d["partition_key_order"]]
        perm = [[str(x) for x in d["out_eps"]].index(k) for k in keys]      # position in topological order of every key
        n = [int(u.out_off[e + 1] - u.out_off[e]) for e in perm]
        shuf.append(UnitArrays(u.in_start, u.in_end, np.concatenate([[0], np.cumsum(n)]),
                               np.concatenate([u.out_start[u.out_off[e]:u.out_off[e + 1]] for e in perm]),
                               np.concatenate([u.out_end[u.out_off[e]:u.out_off[e + 1]] for e in perm]), np.zeros((u.E, u.E))))
        truth.append(d["true_parent"][perm])
        want.append(d["dag"][np.ix_(perm, perm)])
    for got, w in zip(eng.find_order(shuf, truth), want):
        assert np.array_equal(got, w)
    eng.close()


def test_find_order_reproduces_the_reference_dags_emulated(emu_lib):
    check_find_order_on_reference_runs(emu_lib)


@pytest.mark.gpu
def test_find_order_reproduces_the_reference_dags_gpu():
    check_find_order_on_reference_runs(None)


def test_find_order_and_accuracy_emulated(emu_lib):
    check(emu_lib)


@pytest.mark.gpu
def test_find_order_and_accuracy_gpu():
    check(None)
