"""N > 1 path on CPU: two processes (gloo), units partitioned with shard_units, each rank solves its
share (the oracle stands in for the GPU engine here -- this test is about partitioning and the final
gather, the engine itself is covered elsewhere), parents gathered on every rank and compared with the
single-process result."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve(unit):
    import tw_oracle as T

    svc = T.Service(unit.in_start, unit.in_end - unit.in_start, unit.out_off, unit.out_start, unit.out_end - unit.out_start,
                    unit.dag, unit.key_rank)
    end_flag, _, _ = T.windows(svc)
    return T.run_pass(svc, end_flag, gauss=T.gauss_params(svc))["parent"]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (REPO, os.path.join(REPO, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    from traceweaver_amd import sharding, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    units, _ = synth.make_workload(5, 300, services=synth.MEDIA_SERVICES + synth.HOTEL_SERVICES, concurrency=2.0)
    parts = sharding.shard_units([sharding.unit_cost(u) for u in units], world)
    mine = parts[rank]
    local = [_solve(units[k]) for k in mine]
    full = sharding.gather_parents(local, mine, len(units), dist=dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, parts, [p.tolist() for p in full]))


def test_two_rank_partition_and_gather():
    from traceweaver_amd import sharding, synth

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    units, _ = synth.make_workload(5, 300, services=synth.MEDIA_SERVICES + synth.HOTEL_SERVICES, concurrency=2.0)
    want = [_solve(u).tolist() for u in units]
    parts = got[0][1]
    assert sorted(k for p in parts for k in p) == list(range(len(units)))      # a partition: every unit exactly once
    loads = [sum(sharding.unit_cost(units[k]) for k in p) for p in parts]
    assert max(loads) <= 1.5 * min(loads)                                      # reasonably balanced
    for rank, _, full in got:
        assert full == want                                                    # every rank holds the full, identical result


def _e2e_worker(rank, world, port, q, lib_path, corpus_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      TW_TILE="1", TW_COOP_THREADS="1")   # the host-emulation build runs one thread per workgroup
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import torch.distributed as dist

    from traceweaver_amd import sharding
    from traceweaver_amd.engine import Engine
    from traceweaver_amd.ingest import Corpus

    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = Corpus(lib_path=lib_path)
    c.add_directory(corpus_dir, first_span="compose", max_traces=0)
    units, _, n_traces = c.units()
    parts = sharding.shard_units([sharding.unit_cost(u.arrays) for u in units], world)
    mine = [units[k] for k in parts[rank]]
    eng = Engine(0, lib_path=lib_path)
    eng.load([u.arrays for u in mine])
    eng.set_truth([u.true_parent for u in mine], [u.in_trace for u in mine], n_traces)
    eng.run_pass1()
    per, local, flags = eng.evaluate(trace_flags=True)
    total = sharding.end_to_end_accuracy(flags, dist=dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, parts, local, total, [p["correct"] for p in per]))


def test_two_rank_end_to_end_accuracy(emu_lib, tmp_path):
    """Services of one trace solved on different ranks: per-trace accuracy needs the MAX all-reduce of the
    wrong-flags (the one collective of the evaluation path)."""
    from traceweaver_amd import sharding, synth
    from traceweaver_amd.engine import Engine
    from traceweaver_amd.ingest import Corpus

    synth.write_jaeger_corpus(str(tmp_path), 4, 400, app=synth.FANOUT_APP, concurrency=3.0)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_e2e_worker, args=(r, world, port, q, emu_lib, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process, all units on one engine
    c = Corpus(lib_path=emu_lib)
    c.add_directory(str(tmp_path), first_span="compose", max_traces=0)
    units, _, n_traces = c.units()
    eng = Engine(0, lib_path=emu_lib)
    eng.load([u.arrays for u in units])
    eng.set_truth([u.true_parent for u in units], [u.in_trace for u in units], n_traces)
    eng.run_pass1()
    per, (right, right_topk) = eng.evaluate()
    assert len(units) == 2 and got[0][1] == got[1][1] and sorted(k for p in got[0][1] for k in p) == [0, 1]
    for rank, parts, local, total, correct in got:
        assert total == (right, right_topk, n_traces)                 # every rank: the global per-trace figures
        assert correct == [per[k]["correct"] for k in parts[rank]]
        assert local[0] >= right                                      # a rank alone sees fewer wrong traces
    assert right < n_traces                                           # the workload does produce wrong traces


def test_shard_units_is_deterministic_and_complete():
    from traceweaver_amd import sharding

    costs = [5, 1, 9, 3, 3, 7, 2, 8]
    for world in (1, 2, 3, 4, 8):
        parts = sharding.shard_units(costs, world)
        assert parts == sharding.shard_units(costs, world)
        assert sorted(k for p in parts for k in p) == list(range(len(costs)))
    assert sharding.gather_parents([np.zeros((1, 2), np.int32)], [0], 1)[0].shape == (1, 2)


def _run_bench(emu_lib, *flags):
    import json
    import subprocess

    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--lib", emu_lib, "--cpu-sample", "0", "--steps", "1", "--warmup", "1"]
                         + list(flags), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints the one JSON line
    assert out.stdout.rstrip("\n").split("\n")[-1] == lines[0] and len(lines[0]) < 2048, len(lines[0])   # ... last, and short enough for the driver's tail
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks(emu_lib):
    """`python bench.py --gpus 2` launches its own ranks (torch.distributed.run on 127.0.0.1) -- here over gloo against
    the host-emulation build of the engine -- and reports the whole job."""
    one = _run_bench(emu_lib, "--gpus", "1", "--n-in", "1500", "--replicas", "1")
    two = _run_bench(emu_lib, "--gpus", "2", "--backend", "gloo", "--n-in", "1500", "--replicas", "1", "--scale-regimes", "0")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["spans_total"] == 2 * one["config"]["spans_total"] == 2 * two["config"]["spans_per_gpu"]
    assert two["budget_windows"] == 0 and 0.9 < two["accuracy"] <= 1.0
    for key in ("roofline", "gpu_pass_ms", "repaired_windows", "windows"):
        assert key in two


def test_bench_rejects_a_world_size_that_is_not_gpus(emu_lib):
    import subprocess

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--lib", emu_lib, "--gpus", "4", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_bench_alibaba_slice_sharded_over_two_ranks_equals_one(emu_lib):
    """BASELINE.json config 4 in small: one Alibaba-shape slice sharded per service (shard_units on measured work) ->
    engine -> all-gather of the parents; rank 0 re-solves the whole slice alone and the results must be identical."""
    r = _run_bench(emu_lib, "--gpus", "2", "--backend", "gloo", "--workload", "alibaba", "--total-spans", "30000", "--verify", "1")
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["sharded_equals_single_gpu"] is True
    assert sum(r["config"]["spans_per_gpu"]) == r["config"]["spans_total"]
    assert r["budget_windows"] == 0


def _split_worker(rank, world, port, q, lib_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      TW_TILE="1", TW_COOP_THREADS="1")
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import torch.distributed as dist

    from traceweaver_amd import sharding, synth
    from traceweaver_amd.engine import Engine

    dist.init_process_group("gloo", rank=rank, world_size=world)
    units, _ = synth.make_workload(21, 5000, services=["chain3", "par2"], concurrency=1.6)
    parts, psvc, pord, pbase = [], [], [], []
    for s, u in enumerate(units):
        cuts = sharding.split_points(u, world)
        for k, p in enumerate(sharding.split_unit(u, cuts)):
            parts.append(p); psvc.append(s); pord.append(k); pbase.append(([0] + cuts)[k])
    mine = sharding.shard_units([sharding.unit_cost(p) for p in parts], world)[rank]
    eng, fit_eng = Engine(0, lib_path=lib_path), Engine(0, lib_path=lib_path)
    eng.load([parts[k] for k in mine])
    eng.run_pass1()
    p1 = [r["parent"] for r in eng.results(1, fields=("parent",))]
    sharding.refit_split_services(eng, fit_eng, mine, psvc, pord, units, dist=dist)      # all-gather of the gap rows, same fit on every rank
    eng.run_pass2()
    p2 = [r["parent"] for r in eng.results(2, fields=("parent",))]
    full1 = sharding.gather_parents(p1, mine, len(parts), dist=dist)
    full2 = sharding.gather_parents(p2, mine, len(parts), dist=dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, psvc, pbase, [a.tolist() for a in full1], [a.tolist() for a in full2]))


def test_one_service_split_over_two_ranks_equals_the_unsplit_run(emu_lib):
    """Within-service sharding: every service is cut in two at an idle moment on a block boundary, rank r solves its
    parts, the parts' gap rows are all-gathered between the passes and every rank runs the same device refit on the
    union; the stitched parent arrays of both passes equal the single-process, unsplit run bit for bit."""
    from traceweaver_amd import synth
    from traceweaver_amd.engine import Engine

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q, emu_lib)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    units, _ = synth.make_workload(21, 5000, services=["chain3", "par2"], concurrency=1.6)
    eng = Engine(0, lib_path=emu_lib)
    eng.load(units)
    eng.run_pass1()
    want1 = [r["parent"] for r in eng.results(1, fields=("parent",))]
    eng.fit_mixtures(unit_seeds=[0, 1])     # refit_split_services: service s draws from MT19937(seed + s)
    eng.run_pass2()
    want2 = [r["parent"] for r in eng.results(2, fields=("parent",))]
    eng.close()
    _, _, psvc, pbase, _, _ = got[0]
    assert len(psvc) == 4 and sorted(k for g in got for k in g[1]) == [0, 1, 2, 3] and all(len(g[1]) > 0 for g in got)
    for rank, mine, _, _, full1, full2 in got:
        for full, want in ((full1, want1), (full2, want2)):
            for s in range(len(units)):
                sub = [np.array(full[k]) for k in range(len(psvc)) if psvc[k] == s]
                base = [pbase[k] for k in range(len(psvc)) if psvc[k] == s]
                stitched = np.concatenate([np.where(a >= 0, a + b, -1) for a, b in zip(sub, base)], axis=1)
                assert np.array_equal(stitched, want[s])


def test_bench_alibaba_full_matrix_over_two_ranks_equals_one(emu_lib):
    """BASELINE.json config 5 in small: the 15 call graphs at several of exp5's load levels -- services sharded over two ranks,
    every (level, service) pair a unit, uploaded once, all levels resident and scaled on the device by one tw_scale_load call per
    step (the per-service factor of the replica table), solved in one step, parents gathered; rank 0 re-solves the whole matrix
    alone and the results must be identical."""
    r = _run_bench(emu_lib, "--gpus", "2", "--backend", "gloo", "--workload", "alibaba-full", "--total-spans", "24000",
                   "--levels", "1,4000,15000", "--verify", "1")
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["sharded_equals_single_gpu"] is True
    assert r["config"]["spans_total"] == sum(r["config"]["spans_per_gpu"]) and r["config"]["spans_total"] % 3 == 0   # every (level, service) is a unit
    assert list(r["accuracy_by_level"]) == ["1", "4000", "15000"] and all(0.8 < a <= 1.0 for a in r["accuracy_by_level"].values())
    assert r["budget_windows"] == 0


def test_bench_default_line_with_two_ranks_carries_the_sharded_modes(emu_lib):
    """`bench.py --gpus 2` without --workload (what the driver runs for its scaling record): the weak-scaling media line and,
    under `scale_regimes`, BASELINE configs 4 and 5 -- sharded per service, parents all-gathered every step, each verified
    against the single-GPU result on rank 0."""
    import json
    import tempfile

    detail = os.path.join(tempfile.mkdtemp(), "detail.json")
    r = _run_bench(emu_lib, "--gpus", "2", "--backend", "gloo", "--n-in", "600", "--replicas", "1", "--total-spans", "20000", "--levels", "1,4000",
                   "--detail-file", detail)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and "media" in r["config"]["workload"]
    # the full record of the same run: who ran (the process group's world size and backend, every rank's device) ...
    full = json.load(open(detail))
    assert full["ranks"]["world_size"] == 2 and full["ranks"]["backend"] == "gloo" and full["ranks"]["device_of_rank"] == [0, 0]
    assert all(v["ranks"]["world_size"] == 2 for v in full["scale_regimes"].values())
    # ... and the compact line the driver would get from the same record at eight ranks (per-rank span lists of eight entries in the
    # media line and in both sharded modes): still under 2 KB, still carrying the verification of the sharded modes
    sys.path.insert(0, REPO)
    import bench

    full["n_gpus"] = 8
    for rec in [full] + list(full["scale_regimes"].values()):
        per = rec["config"]["spans_per_gpu"]
        rec["config"]["spans_per_gpu"] = (per if isinstance(per, list) else [per, per]) * 4
        rec["n_gpus"] = 8
    line8 = json.dumps(bench.compact_line(full, "gpurun_out/bench_detail_n8.json"), separators=(",", ":"))
    assert len(line8) < 2048, len(line8)
    assert all(v["sharded_equals_single_gpu"] is True for v in json.loads(line8)["scale_regimes"].values())
    regs = r["scale_regimes"]
    assert set(regs) == {"config4_alibaba_slice_sharded", "config5_alibaba_full_sharded"}
    for v in regs.values():
        assert v["scaling"] == "strong" and v["sharded_equals_single_gpu"] is True and v["n_gpus"] == 2 and v["budget_windows"] == 0
        assert v["value"] > 0 and 0.8 < v["accuracy"] <= 1.0 and "frac" in v["roofline"]
    assert list(regs["config5_alibaba_full_sharded"]["accuracy_by_level"]) == ["1", "4000"]


def test_generated_corpus_comes_with_its_replica_table(tmp_path):
    """synth.write_alibaba_corpus(project_root=...) writes data/misc/service_to_replica_new.pickle ({service: [replica ids]},
    executor.py:912) for the services of the corpus: exp5's six compress factors then scale a service by 1 ... 6."""
    import pickle

    from traceweaver_amd import synth, transforms

    d = tmp_path / "data" / "alibaba_microservices" / "call_graph_data" / "call_graph_0"
    synth.write_alibaba_corpus(str(d), 5, 20, project_root=str(tmp_path))
    table = pickle.load(open(tmp_path / "data" / "misc" / "service_to_replica_new.pickle", "rb"))
    assert set(table) == {"gw", "auth", "cart", "catalog", "stock", "db"}
    for f in synth.EXP5_COMPRESS_FACTORS:
        assert all(1 <= transforms.load_factor(f, len(v)) <= 6 for v in table.values())
    assert all(transforms.load_factor(1, len(v)) == 1 for v in table.values())


def test_bench_media_split_over_two_ranks_equals_one(emu_lib):
    """bench.py --workload media-split: six services, each cut in two at an idle moment, gap rows all-gathered between the
    passes, parents gathered at the end; rank 0 re-solves the unsplit services and the stitched result must be identical."""
    r = _run_bench(emu_lib, "--gpus", "2", "--backend", "gloo", "--workload", "media-split", "--n-in", "1200", "--replicas", "2", "--verify", "1")
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["sharded_equals_single_gpu"] is True
    assert sum(r["config"]["spans_per_gpu"]) == r["config"]["spans_total"] and min(r["config"]["spans_per_gpu"]) > 0.3 * r["config"]["spans_total"]


def test_bench_sync_engine_needs_no_torch(emu_lib):
    """`bench.py --sync engine` (what profiles/collect.sh runs under rocprofv3): the same line as the default route,
    from a process that never imports torch (every C-ABI call returns with its stream synchronised)."""
    import subprocess

    a = _run_bench(emu_lib, "--n-in", "600", "--replicas", "1")
    b = _run_bench(emu_lib, "--n-in", "600", "--replicas", "1", "--sync", "engine")
    for k in ("accuracy", "windows", "unassigned", "budget_windows", "n_gpus"):
        assert a[k] == b[k], k
    assert a["config"] == b["config"]
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--lib', %r, '--cpu-sample', '0', '--steps', '1', '--warmup', '0', '--n-in', '300', '--replicas', '1', '--sync', 'engine']\n"
            "runpy.run_path(%r, run_name='__main__')\n"
            "assert 'torch' not in sys.modules, 'torch was imported'\n") % (emu_lib, os.path.join(REPO, "bench.py"))
    env = dict(os.environ, TW_TILE="1", TW_COOP_THREADS="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]


def test_bench_default_batch_is_sixteen_replicas():
    """The default workload of `python bench.py`: 16 replicas of the media graph (profiles/r02d_batch_sweep.json); the other
    workloads keep 4."""
    sys.path.insert(0, REPO)
    import bench

    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        args = bench.parse_args()
        assert args.replicas is None and args.workload == "media" and args.n_in == 100000
    finally:
        sys.argv = old
    import inspect

    src = inspect.getsource(bench.make_units)
    assert "16 if args.workload == \"media\" else 4" in src


def test_cpu_baseline_all_cores_runs_the_sample_once_per_process():
    """bench.py's cpu_baseline: the single-process figure and, next to it, the same sample in several processes at once."""
    sys.path.insert(0, REPO)
    import bench

    old = sys.argv
    try:
        sys.argv = ["bench.py", "--cpu-sample", "300", "--cpu-procs", "2"]
        args = bench.parse_args()
    finally:
        sys.argv = old
    r = bench.cpu_baseline(args, 1000)
    assert r["cores"] == 1 and r["value"] > 0
    a = r["all_cores"]
    assert a["cores"] == 2 and a["value"] > 0 and a["slowest_process_s"] > 0
