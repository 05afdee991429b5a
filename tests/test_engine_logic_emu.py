"""Engine *logic* on the CPU: traceweaver_amd/csrc/tw_engine.hip compiled unchanged with g++ against the
host-emulation shim (tests/hostemu) and compared bit-for-bit with the oracle.  This covers the code
paths that differ structurally from the sequential reference -- parallel prefix scans for the windows,
speculative per-window selection, the consumption repair walk -- without a GPU.  GPU behaviour itself
is covered by tests/test_gpu_parity.py (-m gpu)."""
import os

import numpy as np
import pytest

import parity
from conftest import GOLDEN, golden_mixtures, unit_from_golden


def _golden(name):
    return [f for f in GOLDEN if name in f][0]


@pytest.mark.parametrize("name", ["hotel_load100__search", "nodeio_1__service1", "media_load150__nginx"])
def test_emulated_engine_matches_oracle_on_reference_corpora(emu_lib, name):
    files = [f for f in GOLDEN if name in f]
    if not files:
        pytest.skip("golden %s not present" % name)
    d = np.load(files[0])
    svc, unit = unit_from_golden(d)
    r1, r2, _ = parity.check_units(emu_lib, [unit], mixtures=[golden_mixtures(d)])
    # identical to the frozen reference run except inside windows whose optimum is not unique
    assert (r1[0]["parent"] != d["pass1_parent"]).any(axis=0).sum() <= 4
    assert (r2[0]["parent"] != d["final_parent"]).any(axis=0).sum() <= 4


def test_emulated_engine_matches_oracle_on_stress_units(emu_lib):
    units, _ = parity.stress_units(parity.STRESS)
    r1, r2, _ = parity.check_units(emu_lib, units)
    assert sum(r["repaired_windows"] for r in r1) > 0, "the stress set must exercise the consumption repair walk"


def test_classes_joined_after_the_enumeration(emu_lib, monkeypatch):
    """TW_CLASS_PIPELINE=0: windows and first selection over all tiles on the engine's stream (the route of small batches and of
    skip mode) instead of per endpoint-count class on the class' stream: the same results."""
    monkeypatch.setenv("TW_CLASS_PIPELINE", "0")
    units, _ = parity.stress_units(parity.STRESS)
    r1, r2, _ = parity.check_units(emu_lib, units)
    assert sum(r["repaired_windows"] for r in r1) > 0


@pytest.mark.parametrize("stretches, staged", [(4, "0"), (8, "1000000")])
def test_tile_kernel_in_stretches(emu_lib, monkeypatch, stretches, staged):
    """A class of many tiles is launched in stretches (launch_enumerate): the wavefront kernel takes what a stretch's tiles listed --
    the entries between two snapshots of the list counters -- on a second stream beside the next stretch's tile kernel.  Forced here
    (a tile of the emulated build is one span), staged and joined: the same results."""
    monkeypatch.setenv("TW_STRETCH_MIN_TILES", "1")
    monkeypatch.setenv("TW_ENUM_STRETCHES", str(stretches))
    monkeypatch.setenv("TW_STAGE_MIN_TILES", staged)
    units, _ = parity.stress_units(parity.STRESS)
    r1, r2, _ = parity.check_units(emu_lib, units)
    assert sum(r["repaired_windows"] for r in r1) > 0


def test_selection_takes_every_route_of_the_level_solver(emu_lib):
    """The selection of a component of more than four spans is solved level by level (select_dp): on the small tables of
    k_select_heavy, on the large ones of k_select_dp when a level outgrows those, by the depth-first search when a level
    outgrows those too (the test build's tables hold 6 / 24 states).  The heavy-load units below take all three routes;
    check_units holds every one of them to the oracle's canonical selection."""
    cases = [(2, 400, "chain3", 4, 1), (3, 300, "chain3", 8, 1), (6, 400, "single", 10, 1), (8, 300, "chain2", 12, 1000), (13, 513, "par2", 2, 1000)]
    units, _ = parity.stress_units(cases)
    r1, r2, _ = parity.check_units(emu_lib, units)
    assert sum(r["dp_windows"] for r in r1 + r2) > 0, "no window outgrew the small tables"
    assert sum(r["dfs_components"] for r in r1 + r2) > 0, "no component outgrew the large tables"
    assert sum(r["budget_windows"] for r in r1 + r2) == 0


def test_error_statuses(emu_lib):
    from traceweaver_amd import synth
    from traceweaver_amd.engine import Engine, EngineError, UnitArrays

    u, _ = synth.make_unit(3, 50, shape="chain2")
    eng = Engine(0, lib_path=emu_lib)
    with pytest.raises(EngineError) as ei:       # pass 1 before load
        eng.run_pass1()
    assert ei.value.code == -4
    short = UnitArrays(u.in_start, u.in_end, [0, 49, 99], np.delete(u.out_start, 0), np.delete(u.out_end, 0), u.dag)
    with pytest.raises(EngineError) as ei:       # skip mode (n_out != n_in) is not accelerated
        eng.load([short])
    assert ei.value.code == -2
    one = UnitArrays(u.in_start[:1], u.in_end[:1], [0, 1, 2], u.out_start[[0, 50]], u.out_end[[0, 50]], u.dag)
    with pytest.raises(EngineError) as ei:       # the reference raises on a single incoming span
        eng.load([one])
    assert ei.value.code == -1
    u101, _ = synth.make_unit(4, 101, shape="single")
    eng.load([u101])
    with pytest.raises(EngineError) as ei:       # hazard H3: last block holds one sample -> NaN std
        eng.run_pass1()
    assert ei.value.code == -7
    eng.load([u])
    eng.run_pass1()
    with pytest.raises(EngineError) as ei:       # pass 2 needs mixtures
        eng.run_pass2()
    assert ei.value.code == -4
    g = eng.gaps()
    eng.fit_mixtures()
    mix = eng.mixtures()
    eng.load([u])                                # an engine that only serves the refit (gaps handed over, mixtures fitted) never ran
    eng.set_gaps(g)                              # pass 1 on this batch: the second pass has nothing to read cut-offs and windows from
    eng.set_mixtures([m[0] for m in mix], [m[1] for m in mix])
    with pytest.raises(EngineError) as ei:
        eng.run_pass2()
    assert ei.value.code == -4 and "first pass" in str(ei.value)
    eng.close()


def test_one_shot_entry_point(emu_lib):
    """tw_assign_service (the single-unit convenience of the C-ABI) through raw ctypes."""
    import ctypes

    import tw_oracle as T
    from traceweaver_amd import _ffi, synth

    u, _ = synth.make_unit(51, 300, shape="chain3", concurrency=2)
    lib = _ffi.load(emu_lib)
    h = ctypes.c_void_p()
    assert lib.tw_create(0, ctypes.byref(h)) == 0
    parent = np.empty(u.E * u.n_in, np.int32)
    chosen = np.empty(u.n_in, np.int32)
    res = _ffi.Results(parent.ctypes.data, 0, 0, 0, chosen.ctypes.data, 0, 0, 0)
    rc = lib.tw_assign_service(h, u.n_in, u.in_start.ctypes.data, u.in_end.ctypes.data, u.E, u.out_off.ctypes.data,
                               u.out_start.ctypes.data, u.out_end.ctypes.data, u.dag.ctypes.data, u.key_rank.ctypes.data,
                               None, None, ctypes.byref(res))
    assert rc == 0, lib.tw_last_error(h)
    lib.tw_destroy(h)
    svc = parity.oracle_service(u)
    end_flag, _, _ = T.windows(svc)
    o = T.run_pass(svc, end_flag, gauss=T.gauss_params(svc))
    assert np.array_equal(parent.reshape(u.E, u.n_in), o["parent"]) and np.array_equal(chosen, o["chosen"])


def test_window_width_limit_is_reported(emu_lib):
    """More than 128 candidate spans at one endpoint for one incoming span: TW_ERR_WINDOW_WIDTH, not a wrong answer."""
    from traceweaver_amd.engine import Engine, EngineError, UnitArrays

    n = 200
    in_start = np.arange(n, dtype=np.int64) * 10 + 1_000_000
    in_end = in_start + 100_000                       # every request spans all the others
    out_start = in_start + 5
    out_end = out_start + 3
    u = UnitArrays(in_start, in_end, [0, n], out_start, out_end, [[0]])
    eng = Engine(0, lib_path=emu_lib)
    eng.load([u])
    with pytest.raises(EngineError) as ei:
        eng.run_pass1()
    assert ei.value.code == -5
    eng.close()


def test_engine_reuse_across_batches_of_different_size(emu_lib):
    """One engine, several tw_load_batch calls (the device arena is kept and grown): results must not depend on
    what was loaded before."""
    from traceweaver_amd import synth
    from traceweaver_amd.engine import Engine

    small, _ = parity.stress_units([(51, 120, "par2", 2, 1)])
    big, _ = parity.stress_units([(52, 700, "chain3", 3, 1), (53, 500, "par4", 1.5, 1000)])

    def solve(eng, units):
        eng.load(units)
        eng.run_pass1()
        eng.fit_mixtures()
        eng.run_pass2()
        return eng.results(2)

    fresh_small = solve(Engine(0, lib_path=emu_lib), small)
    fresh_big = solve(Engine(0, lib_path=emu_lib), big)
    eng = Engine(0, lib_path=emu_lib)
    for units, want in ((small, fresh_small), (big, fresh_big), (small, fresh_small), (big, fresh_big)):
        got = solve(eng, units)
        for g, w in zip(got, want):
            for key in ("parent", "topk_idx", "chosen", "leaves", "window_end", "topk_n"):
                assert np.array_equal(g[key], w[key]), key
            assert np.array_equal(g["topk_score"], w["topk_score"], equal_nan=True)
    eng.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_units_match_oracle(emu_lib, seed):
    """Randomised sweep over shapes, load, timestamp granularity and sizes (including sizes around the tile / block
    boundaries): every unit bit-identical to the oracle in both passes."""
    rng = np.random.default_rng(1000 + seed)
    shapes = list(synth_shapes())
    cases = []
    for k in range(7):
        shape = shapes[int(rng.integers(len(shapes)))]
        n = int(rng.choice([2, 3, 57, 99, 100, 127, 128, 129, 200, 255, 256, 257, 301, 402]))
        n += 1 if n % 100 == 1 else 0                       # a last block of one request has no variance (hazard H3)
        wide = len({e for st in synth_shape(shape) for e in st}) >= 6
        conc = float(rng.choice([1.1, 1.5, 2.5] if wide else [1.1, 2.0, 4.0, 7.0, 11.0]))   # wide fan-outs: bounded candidate products
        gran = int(rng.choice([1, 1, 1000]))
        cases.append((seed * 100 + k, n, shape, conc, gran))
    units, _ = parity.stress_units(cases)
    parity.check_units(emu_lib, units)


def synth_shapes():
    from traceweaver_amd import synth
    return synth.SHAPES.keys()


def synth_shape(name):
    from traceweaver_amd import synth
    return synth.SHAPES[name]


def test_lane_threaded_emulation(emu_lib):
    """TW_EMU_LANES=1: workgroups run as host threads, one per lane (64 to a wavefront, 128 per tile, 256 per
    cooperative workgroup), cross-lane operations are rendezvous -- the lane-parallel logic (work-list appends with
    wave-aggregated atomics, ballot prefix sums, block scans, wave reductions, the selection kernels' LDS hand-offs)
    against the oracle, bit for bit -- including the wavefront enumeration kernel (staging by ballot prefix sums, the
    tuple lists of deep call graphs built with wavefront scans, the top-5 bookkeeping) on a 7-endpoint unit."""
    import subprocess
    import sys

    code = (
        "import sys, os\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import parity\n"
        "units, _ = parity.stress_units([(6, 90, 'single', 10, 1), (4, 90, 'par2', 6, 1), (11, 130, 'single', 1.2, 1), (18, 10, 'mix7', 3, 1000)])\n"
        "r1, r2, _ = parity.check_units(%r, units)\n"
        "assert sum(r['repaired_windows'] for r in r1) > 0\n"
        "print('lanes ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
         os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), emu_lib)
    env = dict(os.environ, TW_EMU_LANES="1", TW_TILE="128", TW_COOP_THREADS="256", TW_TILE_SUB="2")   # (two workgroups per tile: the sub-tile launch with a quarter of the host threads of the default eight)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "lanes ok" in out.stdout, out.stderr[-2000:]


def test_lane_threaded_tile_kernel_with_production_tables():
    """k_enumerate_tile by 256 host threads per tile of 128 requests with the table sizes of the HIP build (slices of 192
    outgoing spans, enumerations of up to 256 tuples by the workgroup): block-wide reductions and prefix sums, the slice
    staging, items / tuples spread over the lanes, the rank counting, several segments per tile (par4 at concurrency 3:
    more than 1536 tuples per tile), ordered and unordered call graphs, millisecond-granular ties -- against the oracle."""
    import subprocess
    import sys

    from tests.hostemu.build_emu import build

    code = (
        "import sys, os\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import parity\n"
        "units, _ = parity.stress_units([(41, 140, 'par4', 3, 1), (42, 130, 'chain3', 3, 1), (43, 130, 'par2', 2, 1000), (45, 24, 'mix7', 1.5, 1)])\n"
        "r1, r2, _ = parity.check_units(%r, units, allow_budget=True)\n"
        "print('lanes ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
         os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), build(production=True))
    env = dict(os.environ, TW_EMU_LANES="1", TW_TILE="128", TW_COOP_THREADS="256", TW_TILE_SUB="2")   # (two workgroups per tile: the sub-tile launch with a quarter of the host threads of the default eight)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0 and "lanes ok" in out.stdout, out.stderr[-2000:]


def test_split_enumerations(emu_lib, monkeypatch):
    """Very long enumerations are cut into parts by the first endpoint's candidate and merged (k_merge_parts).  The
    host-emulation build splits from ~100 grid points on: units without call-order constraints (every grid point a
    tuple, the prefix walk) and chains (the tuple list).  A span with two candidates of one endpoint that start together
    is split in log mode (heavy_append in tw_kernels.h); TW_SPLIT_TWINS=0 does not split it, TW_SPLIT_TWINS=1 splits it like the
    others, so that the merge's way back -- order of equal scores not decided: the span is listed again and enumerated whole -- is
    reached (millisecond timestamps).  The five-endpoint chain belongs to a class that defers its long spans: they are cut by their
    listed prefixes once the tuples are counted (list parts, kListSplitFlag in tw_kernels.h).  Every unit equals the oracle bit for bit
    in all three settings (check_units)."""
    from traceweaver_amd.engine import Engine

    cases = [(21, 120, "par4", 4, 1), (22, 120, "chain3", 6, 1), (23, 100, "diamond", 5, 1), (24, 120, "par4", 3, 1000), (25, 80, "chain5", 6, 1),
             (32, 80, "par4", 12, 1), (33, 120, "chain2", 20, 1)]   # the last two: windows wider than 32 candidates (the other instantiation)
    units, _ = parity.stress_units(cases)

    def lists():
        eng = Engine(0, lib_path=emu_lib)
        seen = []
        for u in units:
            eng.load([u])
            eng.run_pass1()
            seen.append(eng.worklists())
        eng.close()
        return seen

    # default (TW_SPLIT_TWINS=2): spans with twin candidates are split in log mode -- every part replays CPython's heap on its share
    # and logs what entered it, k_merge_parts replays the logs; a log that outgrows its dozen entries (test build) sends the span
    # back to be enumerated whole
    parity.check_units(emu_lib, units, allow_budget=True)
    logged = lists()
    assert all(w["split_spans"] > 0 for w in logged), logged
    monkeypatch.setenv("TW_SPLIT_TWINS", "0")   # twins not split: nothing comes back
    parity.check_units(emu_lib, units, allow_budget=True)
    seen = lists()
    assert all(w["split_spans"] > 0 for w in seen), seen
    assert all(w["split_redone"] == 0 for w in seen), seen         # no twins, no undecided order
    assert logged[3]["split_spans"] > seen[3]["split_spans"]        # (the millisecond-granular unit: twins)
    assert logged[3]["split_redone"] < logged[3]["split_spans"] - seen[3]["split_spans"], logged   # ... and most of their logs were complete
    monkeypatch.setenv("TW_SPLIT_TWINS", "1")   # twins split like the others: the order of equal scores is not decided, the span comes back
    parity.check_units(emu_lib, units, allow_budget=True)
    forced = lists()
    assert forced[3]["split_spans"] > seen[3]["split_spans"] and forced[3]["split_redone"] > 0, forced


def test_stress_units_with_production_thresholds(emu_lib):
    """The default host-emulation build uses tiny list thresholds so that small units take every route -- which also means
    that there every wavefront-enumerated span counts as a long one.  Here the same source is built with the thresholds of
    the HIP library: the class' lists hold long, ordinary and wide-window spans side by side, as on the GPU."""
    from tests.hostemu.build_emu import build

    units, _ = parity.stress_units(parity.STRESS)
    parity.check_units(build(production=True), units, allow_budget=True)


def _two_bursts(seed, n_half, shape, concurrency):
    """A service whose requests come in two bursts with an idle moment in between (no request in flight at request n_half)."""
    from traceweaver_amd import synth
    from traceweaver_amd.engine import UnitArrays

    a, _ = synth.make_unit(seed, n_half, shape=shape, concurrency=concurrency)
    b, _ = synth.make_unit(seed + 1, n_half, shape=shape, concurrency=concurrency, t0_us=int(a.in_end.max()) + 10_000_000)
    assert np.array_equal(a.dag, b.dag)
    off = np.arange(a.E + 1, dtype=np.int64) * (2 * n_half)
    os_ = np.concatenate([np.concatenate([a.out_start[a.out_off[e]:a.out_off[e + 1]], b.out_start[b.out_off[e]:b.out_off[e + 1]]]) for e in range(a.E)])
    oe_ = np.concatenate([np.concatenate([a.out_end[a.out_off[e]:a.out_off[e + 1]], b.out_end[b.out_off[e]:b.out_off[e + 1]]]) for e in range(a.E)])
    return UnitArrays(np.concatenate([a.in_start, b.in_start]), np.concatenate([a.in_end, b.in_end]), off, os_, oe_, a.dag, a.key_rank)


# (seeds at which the parts, loaded as whole services, get other windows -- and at 67 / 75 / 62 other assignments -- than the unsplit run)
@pytest.mark.parametrize("seed,shape,conc", [(66, "single", 8.0), (67, "single", 12.0), (75, "single", 12.0), (62, "par2", 6.0), (65, "chain2", 7.0)])
def test_split_service_with_windows_at_the_size_cap(emu_lib, seed, shape, conc):
    """Within-service sharding (sharding.split_unit) when the windows reach the 30-request cap right after the cut: the
    reference's window state machine counts the service's first request twice and never tests its last one for a
    PerfectCut (traceweaver_v3.py:1056-1076); a part that continues after a cut / is followed by one must not repeat that
    (tw_batch.unit_part) -- windows and pass-1 assignments of the stitched parts equal the unsplit run."""
    from traceweaver_amd import sharding
    from traceweaver_amd.engine import Engine

    unit = _two_bursts(seed, 200, shape, conc)
    cuts = sharding.split_points(unit, 2)
    assert cuts == [200]
    parts = sharding.split_unit(unit, cuts)
    assert [p.part for p in parts] == [2, 1]
    eng = Engine(0, lib_path=emu_lib)
    eng.load([unit])
    eng.run_pass1()
    whole = eng.results(1)[0]
    eng.load(parts)
    eng.run_pass1()
    ra, rb = eng.results(1)
    for p in parts:
        p.part = 0
    eng.load(parts)
    eng.run_pass1()
    xa, xb = eng.results(1)
    eng.close()
    assert not np.array_equal(np.concatenate([xa["window_end"], xb["window_end"]]), whole["window_end"])   # the case is one that needs the flags
    assert whole["window_end"][199] == 1 and int(whole["window_end"].sum()) < 100     # long windows on both sides of the cut
    assert np.array_equal(np.concatenate([ra["window_end"], rb["window_end"]]), whole["window_end"])
    stitched = np.concatenate([ra["parent"], np.where(rb["parent"] >= 0, rb["parent"] + 200, -1)], axis=1)
    assert np.array_equal(stitched, whole["parent"])
    assert np.array_equal(np.concatenate([ra["leaves"], rb["leaves"]]), whole["leaves"])


@pytest.mark.parametrize("case", [(71, 57, "chain3", 2.0, 1), (72, 129, "par4", 1.5, 1), (73, 100, "single", 4.0, 1000)])
def test_requests_longer_than_32_bit_offsets(emu_lib, case):
    """k_enumerate_tile stages a tile's candidates as 32-bit offsets from the tile's first start; a request of 2^31 time units or
    more (or that far from the tile's first request) must not be enumerated by it (it goes to the wavefront kernel).  Every timestamp of a stress unit times 5e6 --
    requests of a few hundred microseconds become longer than 2^31 -- and a unit in which only the last request is that
    long: bit-identical to the oracle either way."""
    from traceweaver_amd import synth
    from traceweaver_amd.engine import UnitArrays

    if case[2] not in synth.SHAPES:
        pytest.skip("shape %s not defined" % case[2])
    (u,), _ = parity.stress_units([case])
    k = 5000000
    scaled = UnitArrays(u.in_start * k, u.in_end * k, u.out_off, u.out_start * k, u.out_end * k, u.dag, u.key_rank)
    assert ((scaled.in_end - scaled.in_start) >= 2 ** 31).any()
    in_end = u.in_end.copy()
    in_end[-1] += 3 * 10 ** 9
    tail = UnitArrays(u.in_start, in_end, u.out_off, u.out_start, u.out_end, u.dag, u.key_rank)
    parity.check_units(emu_lib, [scaled, tail, u])
