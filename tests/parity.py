"""Shared engine-vs-oracle comparison used by the host-emulation tier (CPU) and the GPU tier."""
import numpy as np

import tw_oracle as T
from conftest import assert_pass_equal
from traceweaver_amd import synth
from traceweaver_amd.engine import Engine, UnitArrays

# (seed, n_in, shape, concurrency, granularity_us): from lightly interleaved to size-capped windows with
# span consumption across windows, plus millisecond-granular timestamps (exact score ties).
STRESS = [
    (1, 400, "chain3", 1.5, 1), (2, 400, "chain3", 4, 1), (3, 300, "chain3", 8, 1), (4, 300, "par2", 6, 1),
    (5, 300, "diamond", 5, 1), (6, 400, "single", 10, 1), (7, 300, "par4", 3, 1), (8, 300, "chain2", 12, 1000),
    (9, 300, "chain3", 4, 1000), (10, 200, "fan6", 2, 1), (11, 257, "single", 1.2, 1), (12, 2, "chain2", 1, 1),
    (13, 513, "par2", 2, 1000), (14, 150, "chain5", 2.5, 1), (15, 120, "mix7", 1.5, 1), (16, 100, "mix8", 1.3, 1),
    (17, 120, "mix8", 1.2, 1000),
    # deep call graphs with several candidates per endpoint: 10^5-10^6 grid points, up to 2.5e4 feasible tuples per request
    # (the listed-prefix enumeration of k_enumerate_heavy)
    (18, 150, "mix7", 3, 1000), (19, 120, "mix8", 2.5, 1000), (20, 200, "fan6", 3, 1),
]


def oracle_service(u):
    return T.Service(u.in_start, u.in_end - u.in_start, u.out_off, u.out_start, u.out_end - u.out_start, u.dag, u.key_rank,
                     time_scale=getattr(u, "time_scale", None))


def oracle_two_pass(svc, mixtures=None):
    end_flag, pre, win = T.windows(svc)
    p1 = T.run_pass(svc, end_flag, gauss=T.gauss_params(svc))
    if mixtures is None:
        mix_n = np.zeros(svc.nslot, np.int32)
        mix_p = np.zeros((svc.nslot, 5, 3))
        for q, d in enumerate(T.gaps(svc, p1["parent"])):
            if d is not None:
                mix_n[q], mix_p[q] = deterministic_mixture(d)
        mixtures = (mix_n, mix_p)
    p2 = T.run_pass(svc, end_flag, mix_n=mixtures[0], mix_p=mixtures[1])
    return end_flag, p1, p2, mixtures


def deterministic_mixture(d):
    """A cheap, deterministic 1-3 component mixture from quantiles (test input for pass 2 only)."""
    p = np.zeros((5, 3))
    if len(d) == 0:
        return 0, p
    k = 1 if len(np.unique(d)) < 8 else 3
    parts = np.array_split(np.sort(d), k)
    for j, part in enumerate(parts):
        p[j] = (len(part) / len(d), float(part.mean()), 1.0 / np.sqrt(float(part.var()) + 1e-6))
    return k, p


def check_units(lib_path, units, mixtures=None, device=0, allow_budget=False):
    """Runs both passes on the engine for a batch of units and compares every unit with the oracle.

    The engine's and the oracle's exact selection searches return the same selection whenever both complete.  A
    window whose search ran out of its node budget on the engine (counted in budget_windows; tie-saturated inputs
    only) keeps the best selection found: with allow_budget the comparison of such a unit stops at the candidate
    lists of that pass (which do not depend on the selection) and the assignment must still be a valid one;
    without it any such window is a failure."""
    eng = Engine(device, lib_path=lib_path)
    eng.load(units)
    eng.run_pass1()
    r1 = eng.results(1)
    g_eng = eng.gauss_params()
    gaps_eng = eng.gaps()
    ora = []
    for k, u in enumerate(units):
        svc = oracle_service(u)
        ora.append((svc,) + oracle_two_pass(svc, None if mixtures is None else mixtures[k]))
    eng.set_mixtures([o[4][0] for o in ora], [o[4][1] for o in ora])
    eng.run_pass2()
    r2 = eng.results(2)
    eng.close()
    for k, u in enumerate(units):
        svc, end_flag, p1, p2, _ = ora[k]
        tag = "unit %d:" % k
        g = T.gauss_params(svc)
        m = ~np.isnan(g[..., 0])
        assert np.array_equal(g_eng[k][..., 0][m], g[..., 0][m]), tag + " gaussian means"
        assert np.array_equal(g_eng[k][..., 1][m], g[..., 1][m]), tag + " gaussian stds"
        assert np.isnan(g_eng[k][..., 0][~m]).all()
        if r1[k]["budget_windows"] or r2[k]["budget_windows"]:
            assert allow_budget, tag + " %d + %d windows hit the node budget of the selection search" % (r1[k]["budget_windows"], r2[k]["budget_windows"])
            assert p1["budget_windows"] == 0 and p2["budget_windows"] == 0, tag + " the oracle's search did not complete either"
            assert np.array_equal(r1[k]["window_end"], end_flag), tag + " window ends"
            assert np.array_equal(np.transpose(r1[k]["topk_idx"], (2, 0, 1)), p1["topk2_idx"]), tag + " top-5 tuples"
            for r in (r1[k], r2[k]):
                assert_assignment_properties(u, r["parent"])
            assert r1[k]["cnt_unassigned"] >= p1["cnt_unassigned"] if not r1[k]["repaired_windows"] else True
            continue
        assert_pass_equal(r1[k], p1, end_flag, tag + " pass 1")
        for q, go in enumerate(T.gaps(svc, p1["parent"])):
            ge = gaps_eng[k][q]
            ge = ge[~np.isnan(ge)]
            assert (len(ge) == 0) if go is None else np.array_equal(ge, go), tag + " gap samples slot %d" % q
        assert_pass_equal(r2[k], p2, end_flag, tag + " pass 2")
    return r1, r2, ora


def stress_units(cases):
    units, truth = [], []
    for seed, n, shape, conc, gran in cases:
        u, tp = synth.make_unit(seed, n, shape=shape, concurrency=conc, granularity_us=gran)
        units.append(u)
        truth.append(tp)
    return units, truth


def assert_assignment_properties(u, par):
    """Size-independent properties of a parent array [E, n]: all endpoints or none per request, every outgoing
    span used at most once, children inside the parent, call order respected."""
    assigned = par[0] >= 0
    assert ((par >= 0) == assigned).all()
    for e in range(u.E):
        x = par[e][assigned]
        assert len(np.unique(x)) == len(x)
        s = u.out_start[u.out_off[e] + x]
        en = u.out_end[u.out_off[e] + x]
        assert (s >= u.in_start[assigned]).all() and (en <= u.in_end[assigned]).all()
        for p in range(e):
            if u.dag[p, e]:
                assert (u.out_end[u.out_off[p] + par[p][assigned]] <= s).all()


def seeded_chain(lib_path, dataset, goldens, device=0):
    """The whole seeded chain of one frozen reference run (tests/golden/ref_<dataset>__*.npz) through the predictor's
    array route: numpy's global RNG seeded like the run, the services in the run's order (tests/golden/service_order.json),
    per service pass 1 -> the reference's refit on the device, fed exactly the doubles numpy hands scikit-learn at that point
    of the run (incl. the fits the reference discards and its reseeding at the service named "frontend") -> pass 2.
    Yields (golden path, golden, pass-1 leaves, pass-2 result)."""
    import json
    import os

    from traceweaver_amd import skipmode
    from traceweaver_amd.predictor import TraceWeaverGPU

    from conftest import unit_from_golden

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "service_order.json")) as fh:
        order = json.load(fh)[dataset]
    paths = {os.path.basename(p)[len("ref_%s__" % dataset):-4]: p for p in goldens if os.path.basename(p).startswith("ref_%s__" % dataset)}
    assert sorted(paths) == sorted(order), (sorted(paths), order)
    pred = TraceWeaverGPU({}, {}, device=device, fit="device", lib_path=lib_path)
    out = []
    seeded = False
    for svc in order:
        d = np.load(paths[svc])
        if not seeded:
            np.random.seed(int(d["seed"]))
            seeded = True
        _, unit = unit_from_golden(d)
        if svc == "frontend":   # executor.py:1150-1152: create_cache_hits reseeds at every cache rate, also 0
            skipmode.cache_hit_draws(unit.n_in, 0.0)
        r1, r2 = pred.solve_arrays(unit, np.asarray(d["true_parent"]), svc)
        r2["mixtures"] = pred._engine.mixtures()[0]   # (mix_n, mix_p) the device fitted between the passes
        out.append((paths[svc], d, r1, r2))
    pred._engine.close()
    return out


def refit_tie_rows(dataset, service):
    """Mixture rows of a frozen run's service whose fit depends on the order in which binary64 sums are taken -- shown by evaluating
    the restatement of scikit-learn's procedure with its sums taken three ways (tests/golden/make_refit_tie_rows.py): only there may
    the device's mixture table differ from the frozen run's."""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refit_tie_rows.json")) as fh:
        return set(json.load(fh).get(dataset, {}).get(service, []))


def mixture_rows_differing(d, mixtures, rel=1e-6):
    """Slots in which a fitted mixture table differs from the frozen reference run's: another component count, or parameters
    (weight, mean, precision_cholesky) apart by more than `rel`."""
    mn, mp = mixtures
    gn = np.maximum(d["mix_n"], 0).astype(np.int32)
    gp = np.ascontiguousarray(d["mix_p"][:, :, [0, 1, 3]])
    bad = []
    for q in range(len(gn)):
        if gn[q] != mn[q]:
            bad.append(q)
        elif gn[q] > 0:
            a, b = gp[q][:gn[q]], mp[q][:gn[q]]
            if np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300)) > rel:
                bad.append(q)
    return bad


def millisecond_granular(d):
    """True for the frozen runs on millisecond-granular corpora (the nodejs applications).  There mixture components collapse
    onto repeated sample values: the variance of such a component is reg_covar plus the rounding noise of its moments,
    scikit-learn's own result depends on the summation order of its BLAS, and a fit may select another component count on
    another machine: the rows where that happens are listed (refit_tie_rows)."""
    return bool((d["in_start"] % 1000 == 0).all() and (d["out_start"] % 1000 == 0).all() and (d["in_dur"] % 1000 == 0).all())
