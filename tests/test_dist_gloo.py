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


def test_shard_units_is_deterministic_and_complete():
    from traceweaver_amd import sharding

    costs = [5, 1, 9, 3, 3, 7, 2, 8]
    for world in (1, 2, 3, 4, 8):
        parts = sharding.shard_units(costs, world)
        assert parts == sharding.shard_units(costs, world)
        assert sorted(k for p in parts for k in p) == list(range(len(costs)))
    assert sharding.gather_parents([np.zeros((1, 2), np.int32)], [0], 1)[0].shape == (1, 2)
