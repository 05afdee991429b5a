"""GPU tier: the two exchange steps of a sharded run on the engine's own device buffers (sharding.gather_parents_device /
refit_split_services_device: tw_device_buffers -> zero-copy tensor views -> all_gather_into_tensor over RCCL -> joined rows
back into the refit through tw_set_gaps_device) against the host-memory path of the CPU tier's gloo tests.  A gpurun box has
one GPU: the process group is RCCL ("nccl") at world size 1 -- the collective, the views and the device-side joins are the
ones a multi-GPU run uses; only the number of peers differs."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl():
    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_device_views_alias_the_engines_buffers(rccl):
    import torch

    from traceweaver_amd import synth
    from traceweaver_amd.engine import Engine

    units, _ = synth.make_workload(3, 3000, services=["chain3", "par2"], concurrency=2.0)
    eng = Engine(0)
    eng.load(units)
    eng.run_pass1()
    par, gaps = eng.device_views()
    p = torch.as_tensor(par, device="cuda")
    g = torch.as_tensor(gaps, device="cuda")
    host = eng.results(1, fields=("parent",))
    assert np.array_equal(p.cpu().numpy(), np.concatenate([r["parent"].ravel() for r in host]))
    assert np.array_equal(g.cpu().numpy(), np.concatenate([x.ravel() for x in eng.gaps()]), equal_nan=True)
    eng.close()


def test_gather_of_parents_on_device_memory(rccl):
    from traceweaver_amd import sharding, synth
    from traceweaver_amd.engine import Engine

    units, _ = synth.make_workload(5, 4000, services=["chain3", "par2", "single"], concurrency=2.0)
    eng = Engine(0)
    eng.load(units)
    eng.run_pass1()
    eng.fit_mixtures(unit_seeds=[0, 1, 2])
    eng.run_pass2()
    got = sharding.gather_parents_device(eng, [[0, 1, 2]], units, rccl, host_ranks=(0,))
    want = eng.results(2, fields=("parent",))
    for a, b in zip(got, want):
        assert np.array_equal(a, b["parent"])
    eng.close()


def test_split_services_refitted_from_device_memory(rccl):
    """Services cut at idle moments, their parts' gap rows gathered and joined on the device, refitted there: the stitched
    result equals the unsplit run (what tests/test_dist_gloo.py shows for the host-memory path on two ranks)."""
    from traceweaver_amd import sharding, synth
    from traceweaver_amd.engine import Engine

    units, _ = synth.make_workload(21, 5000, services=["chain3", "par2"], concurrency=1.6)
    parts, psvc, pord, pbase = [], [], [], []
    for s, u in enumerate(units):
        cuts = sharding.split_points(u, 2)
        for k, p in enumerate(sharding.split_unit(u, cuts)):
            parts.append(p); psvc.append(s); pord.append(k); pbase.append(([0] + cuts)[k])
    assert len(parts) == 4
    eng, fit_eng = Engine(0), Engine(0)
    eng.load(parts)
    eng.run_pass1()
    sharding.refit_split_services_device(eng, fit_eng, [list(range(len(parts)))], parts, psvc, pord, units, rccl)
    eng.run_pass2()
    got = sharding.gather_parents_device(eng, [list(range(len(parts)))], parts, rccl)
    whole = Engine(0)
    whole.load(units)
    whole.run_pass1()
    whole.fit_mixtures(unit_seeds=[0, 1])
    whole.run_pass2()
    want = [r["parent"] for r in whole.results(2, fields=("parent",))]
    for s in range(len(units)):
        ks = [k for k in range(len(parts)) if psvc[k] == s]
        stitched = np.concatenate([np.where(got[k] >= 0, got[k] + pbase[k], -1) for k in ks], axis=1)
        assert np.array_equal(stitched, want[s])
    for e in (eng, fit_eng, whole):
        e.close()
