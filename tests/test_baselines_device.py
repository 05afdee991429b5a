"""The baseline predictors on the resident table (csrc/tw_baselines.h: FCFS, vPath, WAP5) against
traceweaver_amd/baselines.py -- the restatement that tests/test_baselines.py pins to the reference's own classes -- on
synthetic units of every shape (microsecond- and millisecond-granular, one to eight endpoints), on load-scaled units and on
the frozen inputs of reference runs; WAP5 with its delay samples carried from service to service by callee name."""
import glob
import os

import numpy as np
import pytest

import parity
from conftest import REPO, unit_from_golden
from traceweaver_amd import baselines, synth
from traceweaver_amd.engine import Engine


def host_wap5(units, names):
    w = baselines.WAP5()
    return [baselines.WAP5.parent(u, w.assign(u, n)) for u, n in zip(units, names)]


def check(lib_path, units, truth, names=None):
    names = names or [["svc%d_ep%d" % (k % 3, e) for e in range(u.E)] for k, u in enumerate(units)]   # names repeat across units: shared samples
    eng = Engine(0, lib_path=lib_path)
    eng.load(units)
    eng.set_truth(truth)
    fc = eng.baseline("FCFS")
    vp = eng.baseline("vPath")
    wp, state, opts = eng.wap5(names)
    eng.close()
    w = baselines.WAP5()
    host_opts = [w.assign(u, n) for u, n in zip(units, names)]
    hw = [baselines.WAP5.parent(u, o) for u, o in zip(units, host_opts)]
    assert opts == host_opts
    for k, (u, tp) in enumerate(zip(units, truth)):
        assert np.array_equal(fc[k], baselines.fcfs(u)), "FCFS unit %d" % k
        assert np.array_equal(vp[k], baselines.vpath(u, tp)), "vPath unit %d" % k
        assert np.array_equal(wp[k], hw[k]), "WAP5 unit %d" % k
    return fc, vp, wp


CASES = [(71, 900, "single", 1.5, 1), (72, 700, "par2", 3.0, 1000), (73, 600, "chain3", 2.0, 1), (74, 500, "diamond", 2.5, 1),
         (75, 400, "mix8", 1.5, 1000), (76, 800, "single", 6.0, 1000), (77, 500, "par4", 4.0, 1), (78, 300, "fan6", 2.0, 1)]


def test_device_baselines_match_the_host_restatement(emu_lib):
    units, truth = parity.stress_units(CASES)
    check(emu_lib, units, truth)


def test_device_baselines_on_load_scaled_units(emu_lib):
    from traceweaver_amd import transforms

    units, truth = parity.stress_units(CASES[:4])
    scaled = [transforms.compress_unit(u, tp, f) for u, tp, f in zip(units, truth, (2, 3, 5, 7))]
    check(emu_lib, [s.arrays for s in scaled], [s.true_parent for s in scaled])


def test_device_baselines_on_frozen_reference_inputs(emu_lib):
    paths = [p for p in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "ref_*.npz")))
             if any(k in p for k in ("hotel_load100", "media_load100", "nodeio_1", "node_load150"))]
    ds = [np.load(p) for p in paths]
    units = [unit_from_golden(d)[1] for d in ds]
    truth = [d["true_parent"].astype(np.int32) for d in ds]
    names = [[str(x) for x in d["out_eps"]] for d in ds]
    check(emu_lib, units, truth, names)


def test_skip_mode_batches_are_left_to_the_host(emu_lib):
    from traceweaver_amd import skipmode
    from traceweaver_amd.engine import EngineError

    units, truth = parity.stress_units([(79, 300, "chain3", 1.5, 1)])
    arr, tp, _ = skipmode.cache_hits(units[0], truth[0], 0.2)
    eng = Engine(0, lib_path=emu_lib)
    plan = skipmode.plan(eng, arr)
    eng.load([arr], skip=[plan])
    with pytest.raises(EngineError) as ei:
        eng.baseline("FCFS")
    assert ei.value.code == -2
    eng.close()


@pytest.mark.gpu
def test_device_baselines_on_gpu():
    units, truth = parity.stress_units([(s, n * 20, sh, c, g) for s, n, sh, c, g in CASES])
    fc, vp, wp = check(None, units, truth)
    acc = [float(np.all(p == t, axis=0).mean()) for p, t in zip(vp, truth)]
    assert 0.0 < min(acc) and max(acc) <= 1.0
