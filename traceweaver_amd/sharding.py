"""One process per GPU: independent service units are partitioned across ranks, solved without any
data-path collective, and the parent arrays are gathered once at the end.

(call graph, service) units share nothing but read-only span tables (executor.py:1080-1083 deep-copies
per service), so the only exchange the path needs is the final gather that end-to-end evaluation
(utils.py:99-117 AccuracyEndToEnd / :216-252 ConstructEndToEndTraces) consumes.  `torch.distributed`
with backend "nccl" is RCCL over xGMI on MI355X; payloads are 4 B per outgoing span (KB-MB), i.e.
latency-bound, so a single all_gather of one padded int32 buffer per rank is used.
"""
import numpy as np


def unit_cost(unit, leaves_per_span=None, measured_leaves=None):
    """Work estimate of one unit.  Static: spans x endpoints (what the scans, sorts and per-span kernels cost).
    Enumeration work scales with the candidate product, which spans four orders of magnitude per span and is unknown
    before pass 1: callers that have run a pass give `measured_leaves` (sum of tw_results.leaves of the unit, the tuples
    DfsTraverseX visits, traceweaver_v3.py:302-303), each tuple costing about E score terms."""
    base = unit.n_in * (1 + unit.E)
    if measured_leaves is not None:
        return base + float(measured_leaves) * unit.E
    return base * (leaves_per_span if leaves_per_span else 1.0)


def shard_units(costs, world):
    """Greedy longest-processing-time partition.  Returns `world` lists of unit indices, deterministic."""
    order = sorted(range(len(costs)), key=lambda k: (-costs[k], k))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda j: (load[j], j))
        parts[r].append(k)
        load[r] += costs[k]
    return [sorted(p) for p in parts]


def gather_parents(local_parents, local_ids, n_units, dist=None, device="cpu"):
    """All ranks contribute {unit id: parent [E, n] int32}; every rank returns the full list of
    `n_units` parent arrays.  With dist=None (single process) this is the identity."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = [None] * n_units
        for k, p in zip(local_ids, local_parents):
            out[k] = np.asarray(p, dtype=np.int32)
        return out
    import torch

    world = dist.get_world_size()
    # header: (unit id, E, n) per local unit, then the flattened parents
    head = np.array([[k, p.shape[0], p.shape[1]] for k, p in zip(local_ids, local_parents)], dtype=np.int64).reshape(-1, 3)
    body = np.concatenate([np.asarray(p, dtype=np.int32).ravel() for p in local_parents]) if local_parents else np.zeros(0, np.int32)
    sizes = torch.tensor([head.shape[0], body.shape[0]], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    max_h = max(int(s[0]) for s in all_sizes)
    max_b = max(int(s[1]) for s in all_sizes)
    h = torch.zeros((max_h, 3), dtype=torch.int64, device=device)
    h[: head.shape[0]] = torch.from_numpy(head).to(device)
    bt = torch.zeros(max_b, dtype=torch.int32, device=device)
    bt[: body.shape[0]] = torch.from_numpy(body).to(device)
    hs = [torch.zeros_like(h) for _ in range(world)]
    bs = [torch.zeros_like(bt) for _ in range(world)]
    dist.all_gather(hs, h)
    dist.all_gather(bs, bt)
    out = [None] * n_units
    for r in range(world):
        nh = int(all_sizes[r][0])
        hr = hs[r][:nh].cpu().numpy()
        br = bs[r].cpu().numpy()
        pos = 0
        for k, E, n in hr:
            out[int(k)] = br[pos:pos + int(E) * int(n)].reshape(int(E), int(n)).copy()
            pos += int(E) * int(n)
    return out


def gather_parents_device(engine, rank_units, all_units, dist, host_ranks=(0,)):
    """The final exchange step on the engine's own device buffer (tw_device_buffers): ONE all-gather (ncclAllGather = RCCL over
    xGMI with backend "nccl") of the parent indices as they lie in HBM, 4 B per outgoing span -- no copy through host memory
    on the way in, and the gathered arrays come back to the host only on `host_ranks` (the rank that stitches the
    end-to-end traces).  rank_units[r] = ids of the units rank r holds, in the order it loaded them (the partition every
    rank computes, shard_units); all_units = the units (shapes only).  Returns the list of parent arrays [E, n] per unit on
    the host ranks, None elsewhere."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [sum(all_units[k].E * all_units[k].n_in for k in ids) for ids in rank_units]
    width = max(sizes)
    view, _ = engine.device_views()
    local = torch.as_tensor(view, device="cuda")                       # the engine's buffer, in place
    assert local.numel() == sizes[rank]
    if local.numel() < width:                                            # ranks hold different numbers of spans: pad on the device
        local = torch.cat([local, torch.zeros(width - local.numel(), dtype=torch.int32, device=local.device)])
    out = torch.empty(world * width, dtype=torch.int32, device=local.device)
    dist.all_gather_into_tensor(out, local)
    if host_ranks is not None and rank not in host_ranks:
        return None
    host = out.cpu().numpy()
    res = [None] * len(all_units)
    for r, ids in enumerate(rank_units):
        pos = r * width
        for k in ids:
            u = all_units[k]
            res[k] = host[pos:pos + u.E * u.n_in].reshape(u.E, u.n_in).copy()
            pos += u.E * u.n_in
    return res


def end_to_end_accuracy(trace_flags, dist=None, device="cpu"):
    """AccuracyEndToEnd / TopKAccuracyEndToEnd (helpers/utils.py:99-145) when the services of a trace were solved
    on different ranks: a trace is right iff no rank saw a wrong span of it.  `trace_flags` = the [2, n_traces]
    uint8 wrong-flags of this rank's tw_evaluate (exact / top-5 criterion); one MAX all-reduce (RCCL over xGMI with
    backend "nccl"; n_traces bytes, latency-bound) combines them.  Returns (traces right, traces right under top-5,
    n_traces) on every rank."""
    flags = np.ascontiguousarray(trace_flags, dtype=np.uint8)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch

        t = torch.from_numpy(flags.copy()).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flags = t.cpu().numpy()
    return int((flags[0] == 0).sum()), int((flags[1] == 0).sum()), int(flags.shape[1])


# ------------------------------------------------------------------------------------------------------------------
# One service on several GPUs (BASELINE.json north_star: "(service, time-window) subproblems ...
# all-gather ... to stitch").  What couples the requests of a service in the reference:
#   * windows and span consumption (traceweaver_v3.py:1020-1078, traceweaver_v1.py:457-463): nothing crosses a moment at
#     which no request is in flight -- every earlier request has ended, so has every call it made (containment), the
#     candidate sets on both sides are disjoint (a perfect cut, :1024-1039) and no later request can use an earlier span;
#   * the per-100-request Gaussian parameters of pass 1 (:1173-1178, :580-646): rank slices of the sorted lists -- a cut
#     at a multiple of 100 requests keeps every block whole, and at an idle moment the first i spans of every endpoint
#     list are exactly the calls of the first i requests;
#   * the mixture refit between the passes (:1221-1222): one fit per edge over the gap samples of the whole service --
#     the one exchange step: the parts' gap rows are all-gathered and every rank runs the same device fit on the union
#     (rows joined in request order, the same seeded draws on every rank), then hands the table to its parts.
# So a service is cut at idle moments that fall on block boundaries into parts that are ordinary units; the parts of
# all services are spread over the ranks like whole services are; results are bit-identical to the unsplit run.


def split_points(unit, parts, batch_size=100):
    """Request indices (multiples of batch_size at which no earlier request is still in flight) nearest to the
    equal-size targets; fewer than parts - 1 if the service has too few idle block boundaries."""
    n = unit.n_in
    if parts <= 1 or n < 2 * batch_size:
        return []
    run_max = np.maximum.accumulate(unit.in_end)
    idx = np.arange(batch_size, n - 1, batch_size)
    idx = idx[(run_max[idx - 1] < unit.in_start[idx]) & (idx < n - 1)]
    idx = idx[(n - idx) % batch_size != 1]          # a last block of one request has no variance (hazard H3), in neither part
    cuts = []
    for k in range(1, parts):
        if len(idx) == 0:
            break
        c = int(idx[np.argmin(np.abs(idx - k * n / parts))])
        if c not in cuts and (not cuts or c > cuts[-1]):
            cuts.append(c)
    return cuts


def split_unit(unit, cuts):
    """The parts of `unit` between the cut indices as units of their own (views into the same arrays)."""
    from .engine import UnitArrays

    if unit.time_scale is not None:
        raise ValueError("load-scaled units are split before scaling")
    edges = [0] + list(cuts) + [unit.n_in]
    parts = []
    for a, b in zip(edges[:-1], edges[1:]):
        off = np.arange(unit.E + 1, dtype=np.int64) * (b - a)
        os_ = np.concatenate([unit.out_start[unit.out_off[e] + a:unit.out_off[e] + b] for e in range(unit.E)])
        oe_ = np.concatenate([unit.out_end[unit.out_off[e] + a:unit.out_off[e] + b] for e in range(unit.E)])
        if any(int(unit.out_off[e + 1] - unit.out_off[e]) != unit.n_in for e in range(unit.E)):
            raise ValueError("only no-skip units are split")
        part = (UnitArrays.AFTER_CUT if a > 0 else 0) | (UnitArrays.BEFORE_CUT if b < unit.n_in else 0)
        p = UnitArrays(unit.in_start[a:b], unit.in_end[a:b], off, os_, oe_, unit.dag, unit.key_rank, part=part)
        if a > 0:   # the idle moment: everything before has ended before anything after starts
            assert unit.in_end[:a].max() < unit.in_start[a] and all(
                unit.out_end[unit.out_off[e]:unit.out_off[e] + a].max() < unit.out_start[unit.out_off[e] + a] for e in range(unit.E))
        parts.append(p)
    return parts


def refit_split_services_device(engine, fit_engine, rank_parts, parts, part_service, part_order, service_units, dist, seed=0):
    """refit_split_services on device memory: the gap rows of the local parts are all-gathered from the engine's own buffer
    (tw_device_buffers; ncclAllGather = RCCL over xGMI), the rows of every service are joined in request order on the
    device and handed to the refit without leaving HBM (tw_set_gaps_device); only the fitted tables (a few hundred bytes per
    edge) travel to the host, to be set on the local parts.  rank_parts[r] = part ids on rank r in load order; parts = all
    parts (shapes only)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [sum(parts[p].nslot * parts[p].n_in for p in ids) for ids in rank_parts]
    width = max(sizes)
    _, view = engine.device_views()
    local = torch.as_tensor(view, device="cuda")
    assert local.numel() == sizes[rank]
    if local.numel() < width:
        local = torch.cat([local, torch.zeros(width - local.numel(), dtype=torch.float64, device=local.device)])
    out = torch.empty(world * width, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, local)      # the gather of gap samples: 8 B per (scored edge, request)
    rows = {}
    for r, ids in enumerate(rank_parts):
        pos = r * width
        for p in ids:
            n = parts[p].nslot * parts[p].n_in
            rows[p] = out[pos:pos + n].view(parts[p].nslot, parts[p].n_in)
            pos += n
    services = sorted({part_service[p] for p in rows})
    fit_engine.load([service_units[s] for s in services])
    joined = []
    for s in services:
        mine = sorted((p for p in rows if part_service[p] == s), key=lambda p: part_order[p])
        joined.append(torch.cat([rows[p] for p in mine], dim=1).reshape(-1))
    flat = torch.cat(joined).contiguous()
    torch.cuda.synchronize()
    fit_engine.set_gaps_device(flat.data_ptr())
    fit_engine.fit_mixtures(unit_seeds=[seed + s for s in services])
    tables = {s: (mn.copy(), mp.copy()) for s, (mn, mp) in zip(services, fit_engine.mixtures())}
    local_parts = rank_parts[rank]
    engine.set_mixtures([tables[part_service[p]][0] for p in local_parts], [tables[part_service[p]][1] for p in local_parts])
    return tables


def refit_split_services(engine, fit_engine, local_parts, part_service, part_order, service_units, dist=None, device="cpu", seed=0):
    """The exchange step of split services.  `engine` holds this rank's parts (pass 1 done); local_parts = their global
    part ids, part_service[p] = service of part p, part_order[p] = position of part p inside its service,
    service_units[s] = the unsplit unit of service s (shape and size only).  Gathers the gap rows of all parts on every
    rank, fits every service's edges on the union with the device fit (`fit_engine`, any engine of this rank) and sets
    the tables of the local parts; the draws of service s come from MT19937(seed + s).  Returns {service: (mix_n, mix_p)}."""
    gaps = engine.gaps()
    rows = {p: g for p, g in zip(local_parts, gaps)}
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch

        world = dist.get_world_size()
        head = np.array([[p, g.shape[0], g.shape[1]] for p, g in rows.items()], dtype=np.int64).reshape(-1, 3)
        body = np.concatenate([g.ravel() for g in rows.values()]) if rows else np.zeros(0)
        sizes = torch.tensor([head.shape[0], body.shape[0]], dtype=torch.int64, device=device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        mh, mb = max(int(s[0]) for s in all_sizes), max(int(s[1]) for s in all_sizes)
        h = torch.zeros((mh, 3), dtype=torch.int64, device=device)
        h[:head.shape[0]] = torch.from_numpy(head).to(device)
        bt = torch.zeros(mb, dtype=torch.float64, device=device)
        bt[:body.shape[0]] = torch.from_numpy(body).to(device)
        hs, bs = [torch.zeros_like(h) for _ in range(world)], [torch.zeros_like(bt) for _ in range(world)]
        dist.all_gather(hs, h)
        dist.all_gather(bs, bt)       # the gather of gap samples: 8 B per (scored edge, request), RCCL over xGMI with backend "nccl"
        rows = {}
        for r in range(world):
            hr, br, pos = hs[r][:int(all_sizes[r][0])].cpu().numpy(), bs[r].cpu().numpy(), 0
            for p, a, b in hr:
                rows[int(p)] = br[pos:pos + int(a) * int(b)].reshape(int(a), int(b)).copy()
                pos += int(a) * int(b)
    tables = {}
    services = sorted({part_service[p] for p in rows})
    fit_engine.load([service_units[s] for s in services])
    joined = []
    for s in services:
        mine = sorted((p for p in rows if part_service[p] == s), key=lambda p: part_order[p])
        joined.append(np.concatenate([rows[p] for p in mine], axis=1))
    fit_engine.set_gaps(joined)
    fit_engine.fit_mixtures(unit_seeds=[seed + s for s in services])   # one stream per service: every rank, and the unsplit run, draw the same
    for s, (mn, mp) in zip(services, fit_engine.mixtures()):
        tables[s] = (mn.copy(), mp.copy())
    engine.set_mixtures([tables[part_service[p]][0] for p in local_parts], [tables[part_service[p]][1] for p in local_parts])
    return tables
