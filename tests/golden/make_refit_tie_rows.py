"""Writes tests/golden/refit_tie_rows.json: per frozen reference run and service, the mixture rows (slots) whose fit depends on
the order in which binary64 sums are taken.

The reference refits every scored edge with scikit-learn (ComputeEpPairDistParams5, traceweaver_v3.py:764-786).  On the
millisecond-granular corpora mixture components collapse onto repeated sample values: a component's variance is then reg_covar
plus the rounding noise of its moments, and whether a fit "raises" (covariance <= 0), which component count has the smallest BIC,
and the parameters of the collapsed component depend on the order of the additions -- in scikit-learn itself (its sums are BLAS
reductions).  This script replays every seeded chain (host-emulation build of the engine, the reference's RNG stream), and for every
row evaluates the restatement oracle/tw_refit.py -- pinned to scikit-learn by tests/test_fit.py -- on the row's samples and
uniforms with its M-step sums taken three ways (numpy's pairwise reduction, left to right, correctly rounded): a row on which the
three do not select the same component count with the same parameters (1e-9 relative) is listed.  Only there may the device's
mixture table differ from the frozen run's (tests/test_gpu_parity.py::test_seeded_chain_mixture_tables, tests/test_predictor.py).

    python tests/golden/make_refit_tie_rows.py            (CPU, ~20 minutes)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "oracle")]
os.environ.setdefault("TW_TILE", "1")
os.environ.setdefault("TW_COOP_THREADS", "1")

import tw_refit  # noqa: E402
from conftest import GOLDEN, unit_from_golden  # noqa: E402
from tests.hostemu.build_emu import build  # noqa: E402
from traceweaver_amd import skipmode  # noqa: E402
from traceweaver_amd.engine import Engine  # noqa: E402
from traceweaver_amd.predictor import TraceWeaverGPU, reference_fit_order  # noqa: E402


def fit_modes(x, tape):
    out = []
    for mode in ("pairwise", "sequential", "exact"):
        tw_refit.SUM_MODE = mode
        try:
            out.append(tw_refit.fit_edge(x, tape))
        finally:
            tw_refit.SUM_MODE = "pairwise"
    return out


def sensitive(fits):
    n0, p0 = fits[0]
    for n, p in fits[1:]:
        if n != n0:
            return True
        a, b = p0[:n0], p[:n0]
        if np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300)) > 1e-9:
            return True
    return False


class Recorder(TraceWeaverGPU):
    def _device_refit(self, unit, true_parent):
        eng = self._engine
        if self.replay_true_fit and true_parent is not None:
            self._advance_rng(unit, true_parent)
        max_n = eng.fit_rows()[0]
        off = np.zeros(unit.nslot, dtype=np.int64)
        pos = 0
        order = reference_fit_order(unit, unit.key_rank)
        for q in order:
            off[q] = pos
            pos += Engine.FIT_ROW_DRAWS[int(max_n[q])]
        tape = np.random.random_sample(pos)
        gaps = eng.gaps()[0]
        self.rows = []
        for q in order:
            x = gaps[q][~np.isnan(gaps[q])]
            if len(x) and max_n[q] > 0:
                if sensitive(fit_modes(x, tape[off[q]:off[q] + Engine.FIT_ROW_DRAWS[int(max_n[q])]])):
                    self.rows.append(int(q))
        eng.fit_mixtures(tape=tape, slot_off=[off])


def main():
    lib = build()
    order = json.load(open(os.path.join(HERE, "service_order.json")))
    only = sys.argv[1:]
    path_out = os.path.join(HERE, "refit_tie_rows.json")
    out = json.load(open(path_out)) if only and os.path.exists(path_out) else {}
    for dataset in sorted(order):
        if only and dataset not in only:
            continue
        paths = {os.path.basename(p)[len("ref_%s__" % dataset):-4]: p for p in GOLDEN if os.path.basename(p).startswith("ref_%s__" % dataset)}
        pred = Recorder({}, {}, device=0, fit="device", lib_path=lib)
        seeded = False
        out[dataset] = {}
        for svc in order[dataset]:
            d = np.load(paths[svc])
            if not seeded:
                np.random.seed(int(d["seed"]))
                seeded = True
            _, unit = unit_from_golden(d)
            if svc == "frontend":
                skipmode.cache_hit_draws(unit.n_in, 0.0)
            pred.solve_arrays(unit, np.asarray(d["true_parent"]), svc)
            if pred.rows:
                out[dataset][svc] = pred.rows
                print(dataset, svc, pred.rows, flush=True)
        pred._engine.close()
        out = {k: v for k, v in out.items()}
        with open(path_out, "w") as f:
            json.dump({k: out[k] for k in sorted(out)}, f, indent=1, sort_keys=True)
    print("written", path_out)


if __name__ == "__main__":
    main()
