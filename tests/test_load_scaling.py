"""Load scaling on the resident table (tw_scale_load, traceweaver_amd/csrc/tw_load.h) against the host transform
traceweaver_amd.transforms.compress_unit, which tests/test_transforms.py pins to the reference's own transformed inputs
(tests/golden/refcmp_*.npz, frozen from runs of helpers/transforms.py:10-40): same permutations, same int64 images and
time scales, same rebuilt ground truth, and -- run through both passes -- the same results as a batch uploaded after the
host transform.  CPU tier: host-emulation build; the HIP library under -m gpu."""
import glob
import os

import numpy as np
import pytest

import parity
from conftest import REPO
from traceweaver_amd import transforms
from traceweaver_amd.engine import Engine, EngineError, UnitArrays

CASES = [(1, 400, "chain3", 1.5, 1), (3, 300, "chain3", 8, 1), (5, 300, "diamond", 5, 1), (8, 300, "chain2", 12, 1000),
         (14, 150, "chain5", 2.5, 1), (16, 100, "mix8", 1.3, 1), (7, 300, "par4", 3, 1)]
FACTORS = (3, 1, 2, 1, 3, 7, 2)


def _check(lib, units, truth, factors, trace_keys=None, passes=True):
    host = [transforms.compress_unit(u, tp, f, trace_key=None if trace_keys is None else trace_keys[k])
            for k, (u, tp, f) in enumerate(zip(units, truth, factors))]
    eng = Engine(0, lib_path=lib)
    eng.load(units)
    traces = [np.arange(u.n_in, dtype=np.int32) + 7 * k for k, u in enumerate(units)]
    n_traces = int(max(t.max() for t in traces)) + 1
    eng.set_truth(truth, traces, n_traces)
    ranks = None
    if trace_keys is not None:
        ranks = []
        for key in trace_keys:
            r = np.empty(len(key), dtype=np.int32)
            r[np.argsort(np.asarray(key), kind="stable")] = np.arange(len(key))
            ranks.append(r)
    perms = eng.scale_load(factors, trace_rank=ranks)
    for (ip, ops, ts), s, u in zip(perms, host, units):
        assert ts == s.arrays.time_scale
        assert np.array_equal(ip, s.in_perm)
        for e in range(u.E):
            assert np.array_equal(ops[e], s.out_perm[e]), e
    if not passes:
        return eng, host
    ref = Engine(0, lib_path=lib)
    ref.load([s.arrays for s in host])
    ref.set_truth([s.true_parent for s in host], [t[s.in_perm] for t, s in zip(traces, host)], n_traces)
    for e_ in (eng, ref):
        e_.run_pass1()
    a, b = eng.results(1), ref.results(1)
    for x, y in zip(a, b):
        for k in ("parent", "topk_idx", "topk_n", "chosen", "leaves", "window_end"):
            assert np.array_equal(x[k], y[k]), k
        assert np.array_equal(x["topk_score"], y["topk_score"], equal_nan=True)
    assert eng.evaluate() == ref.evaluate()                     # truth and trace numbers followed the spans
    for x, y in zip(eng.gaps(), ref.gaps()):
        assert np.array_equal(x, y, equal_nan=True)
    for e_ in (eng, ref):
        e_.fit_mixtures(); e_.run_pass2()
    for x, y in zip(eng.results(2), ref.results(2)):
        assert np.array_equal(x["parent"], y["parent"])
    return eng, host


def test_scaled_on_the_device_equals_the_host_transform(emu_lib):
    units, truth = parity.stress_units(CASES)
    _check(emu_lib, units, truth, FACTORS)


def test_load_levels_share_one_upload(emu_lib):
    """exps/exp5 runs six load levels per call graph: every call scales the table as uploaded."""
    units, truth = parity.stress_units(CASES[:3])
    eng = Engine(0, lib_path=emu_lib)
    eng.load(units)
    eng.set_truth(truth)
    for f in (2, 5, 3):
        perms = eng.scale_load([f] * 3)
        eng.run_pass1()
        got = eng.results(1, fields=("parent",))
        host = [transforms.compress_unit(u, tp, f) for u, tp in zip(units, truth)]
        ref = Engine(0, lib_path=emu_lib)
        ref.load([s.arrays for s in host])
        ref.run_pass1()
        for (ip, ops, ts), s, x, y in zip(perms, host, got, ref.results(1, fields=("parent",))):
            assert ts == s.arrays.time_scale and np.array_equal(ip, s.in_perm)
            assert np.array_equal(x["parent"], y["parent"])


def test_trace_id_order_breaks_ties(emu_lib):
    """Spans whose transformed (start, end) coincide keep the order of their trace ids (the reference sorts the partitions
    by trace id first, helpers/transforms.py:13-14, and list.sort is stable)."""
    units, truth = parity.stress_units([(8, 300, "chain2", 4, 1000), (3, 200, "chain3", 3, 1000)])   # ms-granular: many ties
    rng = np.random.default_rng(3)
    keys = [rng.permutation(u.n_in) for u in units]
    eng, host = _check(emu_lib, units, truth, (3, 2), trace_keys=keys)
    plain = [transforms.compress_unit(u, tp, f) for u, tp, f in zip(units, truth, (3, 2))]
    assert any(not np.array_equal(a.in_perm, b.in_perm) for a, b in zip(host, plain))   # the keys did matter


GOLD = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "refcmp_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_reference_corpora(path, emu_lib, oracle):
    """The shipped corpora the reference was run on with --compress_factor 2/3/4: the device transform of the untransformed
    golden inputs gives the int64 image of the float inputs the reference's predictor saw."""
    d = np.load(path)
    corpus, service = str(d["dataset"]).rsplit("_x", 1)[0], str(d["process"])
    base = os.path.join(REPO, "tests", "golden", "ref_%s__%s.npz" % (corpus, service))
    if not os.path.exists(base):
        pytest.skip("no untransformed golden of this service")
    b = np.load(base)
    svc = oracle.service_from_golden(b)
    unit = UnitArrays(svc.in_start, svc.in_end, svc.out_off, svc.out_start, svc.out_end, svc.dag, svc.key_rank)
    eng, host = _check(emu_lib, [unit], [b["true_parent"]], [int(d["compress_factor"])], passes=False)
    assert np.array_equal(host[0].in_start, d["in_start"])      # (the host transform is the reference's, bit for bit)


def test_refusals(emu_lib):
    units, truth = parity.stress_units(CASES[:1])
    eng = Engine(0, lib_path=emu_lib)
    eng.load(units)
    with pytest.raises(EngineError):
        eng.scale_load([2])                                     # no truth: spans of a request cannot be paired
    eng.set_truth(truth)
    with pytest.raises(EngineError):
        eng.scale_load([0])
    bad = [truth[0].copy()]
    bad[0][0, 1] = bad[0][0, 0]                                 # two requests claim the same call
    eng.set_truth(bad)
    with pytest.raises(EngineError):
        eng.scale_load([2])
    s = transforms.compress_unit(units[0], truth[0], 2)
    eng.load([s.arrays])
    eng.set_truth([s.true_parent])
    with pytest.raises(EngineError):
        eng.scale_load([2])                                     # uploaded with a time scale: scaled already


@pytest.mark.gpu
def test_scaled_on_the_gpu_equals_the_host_transform():
    units, truth = parity.stress_units(CASES)
    _check(None, units, truth, FACTORS)
    units, truth = parity.stress_units([(8, 300, "chain2", 4, 1000), (3, 200, "chain3", 3, 1000)])
    rng = np.random.default_rng(3)
    _check(None, units, truth, (3, 2), trace_keys=[rng.permutation(u.n_in) for u in units])


@pytest.mark.gpu
def test_load_levels_at_scale_on_the_gpu():
    """6.4e5 requests per level on the resident table; each level equals the host transform + upload."""
    from traceweaver_amd import synth

    units, truth = synth.make_workload(11, 20000, services=synth.MEDIA_SERVICES, replicas=2, concurrency=1.2)
    eng = Engine(0)
    eng.load(units)
    eng.set_truth(truth)
    for f in (2, 3):
        perms = eng.scale_load([f] * len(units))
        eng.run_pass1()
        got = eng.results(1, fields=("parent", "unit_stats"))
        host = [transforms.compress_unit(u, tp, f) for u, tp in zip(units, truth)]
        ref = Engine(0)
        ref.load([s.arrays for s in host])
        ref.run_pass1()
        for (ip, ops, ts), s, x, y in zip(perms, host, got, ref.results(1, fields=("parent",))):
            assert ts == s.arrays.time_scale and np.array_equal(ip, s.in_perm)
            assert np.array_equal(x["parent"], y["parent"])
