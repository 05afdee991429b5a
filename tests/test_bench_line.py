"""The line `bench.py` prints last is what the driver parses: it must stay a compact JSON object (round 4's 20 KB line was cut
by the driver's 8 KB tail and left the round's headline unparsed)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, REPO)
    import bench

    return bench


def test_compact_line_of_a_full_single_gpu_record_is_under_2kb():
    """profiles/r04j_bench.json is a complete round-4 record (regimes, per-shape reference timings, host legs: 20 KB)."""
    bench = _bench()
    full = json.load(open(os.path.join(REPO, "profiles", "r04j_bench.json")))
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_detail_n1.json"), separators=(",", ":"))
    assert len(line) < 2048, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in back, k
    assert set(back["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "peak_measured"}
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-6
    cb = back["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert cb["reference"]["kind"] == "reference" and cb["reference"]["cores"] == 1 and cb["reference"]["value"] > 0
    assert set(back["regimes_spans_per_s"]) == set(full["regimes"]) and "workload" in back["config"]


def test_compact_line_with_scale_regimes_is_under_2kb():
    bench = _bench()
    full = json.load(open(os.path.join(REPO, "profiles", "r04j_bench.json")))
    full.pop("regimes"); full.pop("cpu_baseline")
    full["n_gpus"] = 8
    full["config"]["spans_per_gpu"] = [25600000] * 8
    reg = {"value": 1.234567e8, "unit": "spans/s", "n_gpus": 8, "steps": 5, "ms_per_step": 12.3456, "scaling": "strong", "accuracy": 0.97654321,
           "budget_windows": 0, "sharded_equals_single_gpu": True, "config": full["config"], "roofline": dict(full["roofline"]),
           "accuracy_by_level": {k: 0.9123456 for k in "1,200,1000,4000,10000,15000".split(",")}}
    full["scale_regimes"] = {"config4_alibaba_slice_sharded": reg, "config5_alibaba_full_sharded": reg}
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_detail_n8.json"), separators=(",", ":"))
    assert len(line) < 2048, len(line)
    back = json.loads(line)
    assert back["scale_regimes"]["config5_alibaba_full_sharded"]["sharded_equals_single_gpu"] is True
