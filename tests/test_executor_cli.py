"""The reference's command line (executor.py:38-74) on the native chain: JSON directory -> ingest -> both passes ->
accuracy -> the reference's result files.  CPU tier through the host-emulation build; the figures are compared
with the frozen reference run of the same corpus: the default refit (`--fit device --seed 10`) is the reference's
procedure on the reference's RNG stream, so the run reproduces the frozen run (np.random.seed(10)) -- per service to
the few requests whose window optimum is proven not unique, end to end exactly but for the traces of those requests
(e2e_band: a listed rule, the millisecond-granular nodejs corpora included)."""
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN

REF = "/root/reference"


def e2e_band(name, traces=1000):
    """Allowed distance of an end-to-end accuracy (percent) from the frozen reference run of corpus `name` -- a listed rule, no
    band: a trace may differ from the frozen run only through a request that lies in a window whose optimum is PROVEN not unique
    (tests/golden/tie_windows.json: north_star's "where the ILP admits ties"), so the distance is at most those requests' share of the
    traces (0 for most corpora, 0.2 - 0.3 pp on the millisecond-granular nodejs corpora whose windows hold such ties); a run with a
    mixture row whose scikit-learn fit is shown to depend on the summation order (tests/golden/refit_tie_rows.json: three runs)
    gets north_star's 0.1 pp on top."""
    import json

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "tie_windows.json")) as f:
        ties = sum(len(v["pass2"]) for k, v in json.load(f).items() if k.startswith("ref_%s__" % name))
    with open(os.path.join(here, "refit_tie_rows.json")) as f:
        rows = any(json.load(f).get(name, {}).values())
    return 100.0 * ties / traces + (0.1 if rows else 0.0) + 1e-9


def run_cli(tmp_path, emu_lib, rel, fix, name):
    from traceweaver_amd import executor

    out = str(tmp_path) + "/"
    argv = ["--relative_path", rel, "--compressed", "0", "--cache_rate", "0", "--fix", str(fix), "--test_name", name,
            "--load_level", "100", "--compress_factor", "1", "--repeat_factor", "1", "--execute_parallel", "0",
            "--results_directory", out, "--clear_cache", "1", "--predictor_indices", "10",
            "--project_root", REF, "--engine_library", emu_lib, "--fit", "device"]
    executor.main(argv)
    suffix = "_%s_100_1_1_0.0.pickle" % name
    return {k: pickle.load(open(out + k + suffix, "rb")) for k in ("accuracy", "process_acc", "confidence_scores", "bin_acc", "e2e")}


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
def test_hotel_run_matches_the_frozen_reference_run(emu_lib, tmp_path):
    got = run_cli(tmp_path, emu_lib, "data/hotel_reservation/hotel_load100/", 2, "hotel_test")
    gold = {str(np.load(p)["process"]): np.load(p) for p in GOLDEN if "hotel_load100__" in p}
    method = "MaxScoreBatchSubsetWithSkips"
    assert set(got["accuracy"]) == {method, method + "TopK"}
    # same services under the same process ids, same request counts
    assert got["process_acc"].keys() == {(method, 0), (method, 1)} and set(got["confidence_scores"]) == set(gold)
    for svc, (acc, not_best, n) in got["confidence_scores"].items():
        g = gold[svc]
        ref_acc = float(np.all(g["final_parent"] == g["true_parent"], axis=0).mean())
        assert n == len(g["in_start"]) and abs(acc - ref_acc) <= 4.0 / n and abs(not_best - int(g["not_best_count"])) <= 4
    ref_e2e = float(gold["frontend"]["e2e_accuracy"])
    assert abs(got["accuracy"][method] - ref_e2e) <= e2e_band("hotel_load100") and got["accuracy"][method + "TopK"] >= got["accuracy"][method]
    assert [p for p, _, _ in got["bin_acc"][method]] == [10.0 * (b + 1) for b in range(10)]
    true_traces, pred_traces = got["e2e"][method]
    assert len(true_traces) == len(pred_traces) == 1000
    tid, spans = next(iter(true_traces.items()))
    assert len(spans) == 5 and all(s[0] == tid for s in spans)      # frontend: 3 calls, search: 2 calls per request


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
@pytest.mark.parametrize("name,rel,fix,n_services", [("media_load100", "data/media_microservices/media_load100/", 1, 6),
                                                     ("nodeio_1", "data/nodejs_microservices_with_arbitrary_file_io/node_1/", 0, 4)])
def test_corpora_that_need_span_surgery(emu_lib, tmp_path, name, rel, fix, n_services):
    """FixSpans2 (media) / FixSpans (nodejs) corpora through the command line: same services as the frozen reference
    run, per-service accuracy within the reference's run-to-run spread of its own figure."""
    got = run_cli(tmp_path, emu_lib, rel, fix, name)
    gold = {str(np.load(p)["process"]): np.load(p) for p in GOLDEN if "ref_%s__" % name in p}
    assert set(got["confidence_scores"]) == set(gold) and len(gold) == n_services
    for svc, (acc, not_best, n) in got["confidence_scores"].items():
        g = gold[svc]
        ref_acc = float(np.all(g["final_parent"] == g["true_parent"], axis=0).mean())
        assert n == len(g["in_start"]) and abs(acc - ref_acc) <= 4.0 / n, (svc, acc, ref_acc)
    ref_e2e = float(next(iter(gold.values()))["e2e_accuracy"])
    assert abs(got["accuracy"]["MaxScoreBatchSubsetWithSkips"] - ref_e2e) <= e2e_band(name)


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
def test_media_load75_directory_ends_where_the_reference_ends(emu_lib, tmp_path):
    """The one shipped corpus the reference cannot finish from its directory: it keeps the first 1001 of media_load75's 1500 traces in
    time order (executor.py:873), and a service of 1001 requests leaves a last parameter block of ONE sample -- std = NaN, the
    reference aborts in its solver (SURVEY.md hazard H3).  The command line loads the same 1001 traces and stops with the status that
    names exactly that: TW_ERR_NAN_PARAMS, no result files, nothing approximated."""
    from traceweaver_amd.engine import EngineError

    with pytest.raises(EngineError) as err:
        run_cli(tmp_path, emu_lib, "data/media_microservices/media_load75/", 1, "media_load75")
    assert err.value.code == -7 and "TW_ERR_NAN_PARAMS" in str(err.value)
    assert not [f for f in os.listdir(str(tmp_path)) if f.endswith(".pickle")]


def _all_corpora():
    from test_ingest import REFERENCE_CORPORA
    # (media_load75 is frozen on its first 1000 files by name; the command line reads a directory, where the reference keeps the
    # first 1001 of the 1500 traces in time order and then fails on a NaN parameter block, hazard H3: array-level tests only)
    return [c for c in REFERENCE_CORPORA if c[0] != "media_load75"]


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
@pytest.mark.parametrize("name,rel,fix", _all_corpora(), ids=[c[0] for c in _all_corpora()])
def test_end_to_end_accuracy_tracks_the_reference_on_every_corpus(emu_lib, tmp_path, name, rel, fix):
    got = run_cli(tmp_path, emu_lib, "data/" + rel + "/", fix, name)
    g = np.load([p for p in GOLDEN if "ref_%s__" % name in p][0])
    method = "MaxScoreBatchSubsetWithSkips"
    assert int(g["seed"]) == 10   # run_cli's default --seed
    assert abs(got["accuracy"][method] - float(g["e2e_accuracy"])) <= e2e_band(name)
    assert abs(got["accuracy"][method + "TopK"] - float(g["e2e_topk_accuracy"])) <= 0.1


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
@pytest.mark.parametrize("name,rel,fix", [c for c in _all_corpora() if c[0] in ("hotel_load100", "hotel_load150", "media_load100", "node_load150",
                                                                                "nodeio_0.2", "nodeio_1")],
                         ids=["hotel_load100", "hotel_load150", "media_load100", "node_load150", "nodeio_0.2", "nodeio_1"])
def test_fit_sklearn_reproduces_the_seeded_reference_run(emu_lib, tmp_path, name, rel, fix):
    """--fit sklearn --seed 10 (the cross-check of the default): scikit-learn itself on the host where the device refit
    runs otherwise, the RNG stream replayed service by service -- the command line reproduces the frozen reference run
    (np.random.seed(10)) of the corpus: per-service accuracy to within the few requests whose window optimum is not unique,
    end-to-end accuracy to +-0.1 pp (+-0.25 pp on the millisecond-granular corpora; SURVEY.md hazard H9)."""
    from traceweaver_amd import executor

    out = str(tmp_path) + "/"
    executor.main(["--relative_path", "data/" + rel + "/", "--compressed", "0", "--cache_rate", "0", "--fix", str(fix), "--test_name", name,
                   "--load_level", "100", "--compress_factor", "1", "--repeat_factor", "1", "--execute_parallel", "0", "--results_directory", out,
                   "--clear_cache", "1", "--predictor_indices", "10", "--project_root", REF, "--engine_library", emu_lib, "--fit", "sklearn",
                   "--seed", "10"])
    suffix = "_%s_100_1_1_0.0.pickle" % name
    acc = pickle.load(open(out + "accuracy" + suffix, "rb"))
    conf = pickle.load(open(out + "confidence_scores" + suffix, "rb"))
    gold = {str(np.load(p)["process"]): np.load(p) for p in GOLDEN if "ref_%s__" % name in p}
    assert set(conf) == set(gold)
    for svc, (a, not_best, n) in conf.items():
        g = gold[svc]
        ref = float(np.all(g["final_parent"] == g["true_parent"], axis=0).mean())
        assert abs(a - ref) <= 4.0 / n, svc
    g0 = next(iter(gold.values()))
    method = "MaxScoreBatchSubsetWithSkips"
    assert abs(acc[method] - float(g0["e2e_accuracy"])) <= e2e_band(name)
    assert abs(acc[method + "TopK"] - float(g0["e2e_topk_accuracy"])) <= e2e_band(name)


def _band():
    import json
    return json.load(open(os.path.join(os.path.dirname(GOLDEN[0]), "ref_accuracy_band.json")))


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
@pytest.mark.parametrize("name,seed_pos", [("hotel_load100", 3), ("nodeio_1", 2), ("nodeio_1", 1), ("node_load150", 3), ("media_load125", 4),
                                           ("media_load150", 0), ("media_load150", 3)])
def test_seeded_runs_sit_in_the_references_multi_seed_band(emu_lib, tmp_path, name, seed_pos):
    """tests/golden/ref_accuracy_band.json holds the end-to-end accuracy of the unmodified reference for five values of
    np.random.seed per corpus (oracle/refrun/gen_golden.py --band): its refit draws from the global RNG (hazard H9), so its
    own figure moves by up to 4 pp between seeds.  The command line with the same seed lands on the same figure (the
    extremes of the band included) -- with the device refit (default) and with scikit-learn on the host.  The batched mode
    draws from another stream (the engine's MT19937, one block per slot): one more sample of the same spread."""
    band = _band()[name]
    rel, fix = next((c[1], c[2]) for c in _all_corpora() if c[0] == name)
    from traceweaver_amd import executor

    def run(fit, seed):
        out = str(tmp_path) + "/%s_%d/" % (fit, seed)
        os.makedirs(out)
        executor.main(["--relative_path", "data/" + rel + "/", "--compressed", "0", "--cache_rate", "0", "--fix", str(fix), "--test_name", name,
                       "--load_level", "100", "--compress_factor", "1", "--repeat_factor", "1", "--execute_parallel", "0",
                       "--results_directory", out, "--clear_cache", "1", "--predictor_indices", "10", "--project_root", REF,
                       "--engine_library", emu_lib, "--fit", fit, "--seed", str(seed)])
        return pickle.load(open(out + "accuracy_%s_100_1_1_0.0.pickle" % name, "rb"))

    method = "MaxScoreBatchSubsetWithSkips"
    for fit in ("device", "sklearn") if seed_pos in (1, 3) else ("device",):
        acc = run(fit, band["seeds"][seed_pos])
        assert abs(acc[method] - band["e2e"][seed_pos]) <= e2e_band(name), fit
        assert abs(acc[method + "TopK"] - band["e2e_topk"][seed_pos]) <= e2e_band(name), fit
    if seed_pos == 3:
        spread = max(band["e2e"]) - min(band["e2e"])
        dev = run("device-batch", 0)
        assert min(band["e2e"]) - max(spread, 0.5) <= dev[method] <= max(band["e2e"]) + max(spread, 0.5)


def test_unsupported_settings_are_refused(emu_lib, tmp_path):
    from traceweaver_amd import executor

    base = ["--relative_path", "x", "--fix", "2", "--results_directory", str(tmp_path) + "/", "--engine_library", emu_lib]
    for extra in (["--cache_rate", "0", "--predictor_indices", "2,10"], ["--cache_rate", "0", "--parallel", "1"]):
        with pytest.raises(SystemExit) as ei:
            executor.main(base + extra)
        assert "not supported here" in str(ei.value)


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
@pytest.mark.parametrize("rate", ["0.1", "0.3"])
def test_cache_hit_runs_of_exp2(emu_lib, tmp_path, rate):
    """exps/exp2/run_experiment.sh: hotel_load150 with --cache_rate R and predictors "3,4,10": cache hits are injected into
    `frontend` (skip mode, one pass with skip spans), `search` runs the two passes; the figures against the reference run
    with the same rate (tests/golden/refskip_*)."""
    import glob

    from traceweaver_amd import executor

    out = str(tmp_path) + "/"
    executor.main(["--relative_path", "data/hotel_reservation/hotel_load150/", "--compressed", "0", "--cache_rate", rate, "--fix", "2",
                   "--test_name", "x", "--load_level", "150", "--compress_factor", "1", "--repeat_factor", "1", "--execute_parallel", "0",
                   "--results_directory", out, "--clear_cache", "1", "--predictor_indices", "3,4,10", "--project_root", REF,
                   "--engine_library", emu_lib])
    suffix = "_x_150_1_1_%s.pickle" % float(rate)
    acc = pickle.load(open(out + "accuracy" + suffix, "rb"))
    conf = pickle.load(open(out + "confidence_scores" + suffix, "rb"))
    gold = {str(np.load(p)["process"]): np.load(p) for p in glob.glob(os.path.join(os.path.dirname(GOLDEN[0]), "refskip_hotel_load150_c%s__*.npz" % rate.replace(".", "p")))}
    g = gold["frontend"]
    ref_acc = float(np.all(g["final_parent"] == g["true_parent"], axis=0).mean())
    assert abs(conf["frontend"][0] - ref_acc) <= 0.005 and conf["frontend"][2] == 1000
    from conftest import skip_tie_requests

    assert abs(conf["frontend"][1] - int(g["not_best_count"])) <= len(skip_tie_requests("refskip_hotel_load150_c%s__frontend" % rate.replace(".", "p")))
    method = "MaxScoreBatchSubsetWithSkips"
    assert abs(acc[method] - float(g["e2e_accuracy"])) < 1.5 and acc[method + "TopK"] >= acc[method]
    assert acc["FCFS"] < acc[method] and acc["WAP5"] < acc[method]


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
def test_baseline_columns(emu_lib, tmp_path):
    """--predictor_indices "3,4,7,10" (what exps/exp1 asks for): WAP5, FCFS and vPath columns next to the accelerated
    predictor, one set of files."""
    from traceweaver_amd import executor

    out = str(tmp_path) + "/"
    executor.main(["--relative_path", "data/hotel_reservation/hotel_load100/", "--cache_rate", "0", "--fix", "2", "--test_name", "b",
                   "--load_level", "100", "--results_directory", out, "--predictor_indices", "3,4,7,10", "--project_root", REF,
                   "--engine_library", emu_lib])   # exps/exp1/run_experiment.sh:40
    acc = pickle.load(open(out + "accuracy_b_100_1_1_0.0.pickle", "rb"))
    proc = pickle.load(open(out + "process_acc_b_100_1_1_0.0.pickle", "rb"))
    assert list(acc) == ["WAP5", "FCFS", "vPath", "MaxScoreBatchSubsetWithSkips", "MaxScoreBatchSubsetWithSkipsTopK"]
    assert all(acc["MaxScoreBatchSubsetWithSkips"] > acc[m] for m in ("WAP5", "FCFS", "vPath"))
    assert {k[0] for k in proc} == {"WAP5", "FCFS", "vPath", "MaxScoreBatchSubsetWithSkips"} and len(proc) == 8


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
def test_load_scaling_through_the_command_line(emu_lib, tmp_path):
    """--compress_factor 2 on hotel_load50 with one replica per service: the units the engine solves are the inputs the
    reference's predictor saw in the frozen run of the same command (tests/golden/refcmp_hotel_load50_x2__*), pass 1
    agrees with it request for request, and the end result stays at the pass-1 level -- the reference's own second pass
    ends at 0 % on this path (traceweaver_amd/transforms.py)."""
    import glob

    from conftest import REPO
    from traceweaver_amd import executor

    out = str(tmp_path) + "/"
    replicas = str(tmp_path / "replicas.pickle")
    with open(replicas, "wb") as f:
        pickle.dump({"frontend": [0], "search": [0]}, f)
    argv = ["--relative_path", "data/hotel_reservation/hotel_load50/", "--compressed", "0", "--cache_rate", "0", "--fix", "2",
            "--test_name", "hotel_x2", "--load_level", "50", "--compress_factor", "2", "--repeat_factor", "1", "--execute_parallel", "0",
            "--results_directory", out, "--clear_cache", "1", "--predictor_indices", "10", "--project_root", REF,
            "--engine_library", emu_lib, "--replicas_file", replicas]
    executor.main(argv)
    acc = pickle.load(open(out + "accuracy_hotel_x2_50_2_1_0.0.pickle", "rb"))
    conf = pickle.load(open(out + "confidence_scores_hotel_x2_50_2_1_0.0.pickle", "rb"))
    gold = {str(np.load(p)["process"]): np.load(p) for p in glob.glob(os.path.join(REPO, "tests", "golden", "refcmp_hotel_load50_x2__*.npz"))}
    assert set(conf) == set(gold) == {"frontend", "search"}
    for svc, (a, _, n) in conf.items():
        g = gold[svc]
        ref_pass1 = float(np.all(g["pass1_parent"] == g["true_parent"], axis=0).mean())
        ref_final = float(np.all(g["final_parent"] == g["true_parent"], axis=0).mean())
        assert n == 1000 and a >= ref_pass1 - 0.02 and ref_final < 0.05
    assert acc["MaxScoreBatchSubsetWithSkips"] > 95.0 and float(gold["frontend"]["e2e_accuracy"]) == 0.0
    with pytest.raises(SystemExit) as ei:                       # no replica table: says which file it wants
        executor.main(argv[:-1] + [str(tmp_path / "missing.pickle")])
    assert "needs the replica table" in str(ei.value)


def _alibaba_shape_with_load_scaling(lib, tmp_path):
    """The exps/exp5 route on a generated corpus of the Alibaba parser's shape: --fix 5 --compress_factor 3 with a
    replica table; the self-call stand-in takes the replica count of the service it was split from."""
    from traceweaver_amd import executor, synth

    synth.write_alibaba_corpus(str(tmp_path / "call_graph_0"), 21, 700, concurrency=1.2, violations=0.02)
    out = str(tmp_path / "res") + "/"
    replicas = str(tmp_path / "replicas.pickle")
    with open(replicas, "wb") as f:
        pickle.dump({"gw": [0], "cart": [0, 1], "catalog": [0, 1, 2]}, f)
    argv = ["--absolute_path", str(tmp_path / "call_graph_0"), "--cache_rate", "0", "--fix", "5", "--results_directory", out,
            "--test_name", "cg0", "--load_level", "0", "--compress_factor", "3", "--replicas_file", replicas,
            "--predictor_indices", "3,4,7,10"] + (["--engine_library", lib] if lib else [])   # exps/exp5/run_experiment.sh:58
    acc, per_process, conf = executor.run(executor.parse_args(argv))
    plain = executor.run(executor.parse_args([a if a != "3" else "1" for a in argv]))[0]
    assert os.path.exists(out + "accuracy_cg0_0_3_1_0.0.pickle") and os.path.exists(out + "accuracy_cg0_0_1_1_0.0.pickle")
    assert len(conf) == 4 and sum(k.endswith("-loop") for k in conf) == 1
    method = "MaxScoreBatchSubsetWithSkips"
    assert plain[method] > 90.0 and acc[method] > 90.0 and acc[method + "TopK"] >= acc[method]
    assert list(acc) == ["WAP5", "FCFS", "vPath", method, method + "TopK"] and all(acc[method] > acc[m] for m in ("WAP5", "FCFS", "vPath"))


def test_alibaba_shape_with_load_scaling(emu_lib, tmp_path):
    _alibaba_shape_with_load_scaling(emu_lib, tmp_path)


@pytest.mark.gpu
def test_alibaba_shape_with_load_scaling_gpu(tmp_path):
    _alibaba_shape_with_load_scaling(None, tmp_path)


def test_generated_corpus_end_to_end(emu_lib, tmp_path):
    """No reference data needed: a generated plain-Jaeger corpus through the same command line."""
    from traceweaver_amd import executor, synth

    synth.write_jaeger_corpus(str(tmp_path / "corpus"), 11, 400, app=synth.HOTEL_APP, concurrency=1.5)
    out = str(tmp_path / "res") + "/"
    executor.main(["--absolute_path", str(tmp_path / "corpus"), "--cache_rate", "0", "--fix", "2", "--results_directory", out,
                   "--test_name", "gen", "--load_level", "7", "--engine_library", emu_lib])
    acc = pickle.load(open(out + "accuracy_gen_7_1_1_0.0.pickle", "rb"))
    assert acc["MaxScoreBatchSubsetWithSkips"] > 85.0


@pytest.mark.skipif(not os.path.isdir(REF + "/data"), reason="the reference's data directory is not present")
def test_reference_path_shim(emu_lib, tmp_path):
    """The path and the argument list exps/exp1/run_experiment.sh uses (lines 5-18, 40-46), as a subprocess."""
    import subprocess
    import sys

    from conftest import REPO

    out = str(tmp_path) + "/"
    env = dict(os.environ, TRACEWEAVER_ROOT=REF, TW_TILE="1", TW_COOP_THREADS="1")
    cmd = [sys.executable, os.path.join(REPO, "src/trace_reconstructor/ports/python/executor.py"),
           "--relative_path", "data/hotel_reservation/hotel_load25/", "--compressed", "0", "--cache_rate", "0", "--fix", "2",
           "--test_name", "hotel_test", "--load_level", "25", "--compress_factor", "1", "--repeat_factor", "1",
           "--execute_parallel", "0", "--results_directory", out, "--clear_cache", "1", "--predictor_indices", "3,4,7,10",
           "--engine_library", emu_lib]
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, timeout=300)
    acc = pickle.load(open(out + "accuracy_hotel_test_25_1_1_0.0.pickle", "rb"))
    assert set(acc) == {"WAP5", "FCFS", "vPath", "MaxScoreBatchSubsetWithSkips", "MaxScoreBatchSubsetWithSkipsTopK"}
    assert acc["MaxScoreBatchSubsetWithSkips"] == 100.0       # the frozen reference run scored 100.0 here too
