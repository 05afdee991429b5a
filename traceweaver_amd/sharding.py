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
