"""Synthetic service units in structure-of-arrays form (seed-fixed, integer microseconds).

The shipped Jaeger corpora of the reference hold 1000 requests per application and the Alibaba corpus
is a Git-LFS pointer, so throughput is measured on synthetic units that keep the
*shape* of the reference's inputs: one incoming span per request, exactly one outgoing call per
(request, endpoint) (utils.py:22-32 assumes that), children contained in the parent, and a call-order
DAG obtained exactly like executor.py:214-285 (FindOrder): a->b iff a finished before b started in
every request.

`granularity_us=1000` reproduces the millisecond-granular timestamps of the nodejs and Alibaba-shape
data (alibaba-analysis/real-parser.py:323,329), where exact score ties do occur.
"""
import numpy as np

from .engine import UnitArrays

# call structures: list of stages, every stage is a tuple of endpoints called in parallel; stages run
# one after the other.  Endpoint ids are arbitrary labels.
SHAPES = {
    "single": [(0,)],
    "chain2": [(0,), (1,)],
    "chain3": [(0,), (1,), (2,)],                 # hotel frontend: search -> reservation -> profile
    "par2": [(0, 1)],                             # nodejs service1
    "par4": [(0, 1, 2, 3)],                       # media nginx: four parallel calls
    "diamond": [(0,), (1, 2), (3,)],
    "fan6": [(0,), (1, 2, 3, 4), (5,)],
    "chain5": [(0,), (1,), (2,), (3,), (4,)],
    "mix7": [(0, 1), (2,), (3, 4, 5), (6,)],
    "mix8": [(0,), (1, 2, 3), (4, 5), (6,), (7,)],   # TW_MAX_EP endpoints
}

# the six accelerated services of media_microservices (E in {1,1,1,1,2,4})
MEDIA_SERVICES = ["single", "single", "single", "single", "par2", "par4"]
HOTEL_SERVICES = ["chain3", "chain2"]


def find_order(start, end):
    """executor.py:214-285 on ground-truth arrays [E, n]: dag[a, b] = 1 iff end[a] <= start[b] in every request."""
    E = start.shape[0]
    dag = np.zeros((E, E), dtype=np.uint8)
    for a in range(E):
        for b in range(E):
            if a != b and bool(np.all(end[a] <= start[b])):
                dag[a, b] = 1
    return dag


def topo_order(dag):
    """A topological order with ties in index order (what networkx returns for insertion-ordered nodes)."""
    E = dag.shape[0]
    indeg = dag.sum(axis=0).astype(int)
    order, ready = [], [e for e in range(E) if indeg[e] == 0]
    while ready:
        e = ready.pop(0)
        order.append(e)
        for f in range(E):
            if dag[e, f]:
                indeg[f] -= 1
                if indeg[f] == 0:
                    ready.append(f)
    if len(order) != E:
        raise ValueError("call-order relation is not a DAG")
    return order


def make_unit(seed, n_in, shape="chain3", concurrency=1.5, mean_service_us=4000.0, sigma=0.5,
              gap_us=300.0, granularity_us=1, t0_us=1_600_000_000_000_000):
    """Returns (UnitArrays, true_parent [E, n_in] int32).

    concurrency ~ mean number of requests in flight (arrival rate x mean response time)."""
    rng = np.random.default_rng(seed)
    stages = SHAPES[shape] if isinstance(shape, str) else shape
    eps = sorted({e for st in stages for e in st})
    E = len(eps)
    n = int(n_in)
    q = int(granularity_us)

    def ln(mean, size):
        mu = np.log(mean) - 0.5 * sigma * sigma
        return np.maximum(1, rng.lognormal(mu, sigma, size)).astype(np.int64)

    start = np.zeros((E, n), dtype=np.int64)
    end = np.zeros((E, n), dtype=np.int64)
    t = ln(gap_us, n)  # offset of the first stage relative to the request start
    for st in stages:
        stage_end = np.zeros(n, dtype=np.int64)
        for e in st:
            s = t + ln(gap_us, n) // 4
            d = ln(mean_service_us, n)
            start[e], end[e] = s, s + d
            stage_end = np.maximum(stage_end, s + d)
        t = stage_end + ln(gap_us, n)
    resp = t  # response time of the request
    mean_resp = float(resp.mean())
    inter = rng.exponential(mean_resp / max(concurrency, 1e-9), n)
    in_start = t0_us + np.cumsum(inter).astype(np.int64)
    in_end = in_start + resp
    start += in_start
    end += in_start
    if q > 1:  # quantise, keeping containment and stage order
        in_start = in_start // q * q
        start = start // q * q
        end = -(-end // q) * q
        end = np.maximum(end, start)
        for k in range(1, len(stages)):  # a later stage must not start before the previous stage ended
            prev_end = np.max(end[list(stages[k - 1])], axis=0)
            for e in stages[k]:
                shift = np.maximum(0, prev_end - start[e])
                start[e] += shift
                end[e] += shift
        in_end = np.maximum(-(-in_end // q) * q, end.max(axis=0))
    dag_true = find_order(start, end)
    order = topo_order(dag_true)
    dag = dag_true[np.ix_(order, order)]
    # sort every list by (start, end) and remember where each request's span went
    in_perm = np.lexsort((in_end, in_start))
    in_rank = np.empty(n, dtype=np.int64)
    in_rank[in_perm] = np.arange(n)
    out_start, out_end, true_parent = [], [], np.zeros((E, n), dtype=np.int32)
    for k, e in enumerate(order):
        perm = np.lexsort((end[e], start[e]))
        rank = np.empty(n, dtype=np.int64)
        rank[perm] = np.arange(n)
        out_start.append(start[e][perm])
        out_end.append(end[e][perm])
        true_parent[k, in_rank] = rank
    unit = UnitArrays(in_start[in_perm], in_end[in_perm], np.arange(E + 1, dtype=np.int64) * n,
                      np.concatenate(out_start), np.concatenate(out_end), dag)
    return unit, true_parent


def make_workload(seed, n_in_per_unit, services=MEDIA_SERVICES, replicas=1, **kw):
    """A batch of independent units: `replicas` copies of the given service list (different seeds)."""
    units, truth = [], []
    for r in range(replicas):
        for k, shape in enumerate(services):
            u, tp = make_unit(seed * 1000003 + r * 101 + k, n_in_per_unit, shape=shape, **kw)
            units.append(u)
            truth.append(tp)
    return units, truth


def accuracy(parent, true_parent):
    """utils.py:62-79 AccuracyForService on index arrays: a request counts when every endpoint matches."""
    return float(np.all(parent == true_parent, axis=0).mean())


# ---------------------------------------------------------------------------------------------
# Jaeger-JSON corpora (one trace per file, the shape of the reference's data/*/ directories): a small
# application as a call tree -- every service handles a request with a server span and calls its callees
# through client spans, stage after stage.
HOTEL_APP = {"root_op": "HTTP GET /hotels", "root": "frontend",
             "calls": {"frontend": [("search",), ("reservation",), ("profile",)], "search": [("geo",), ("rate",)]}}
# media_microservices shape: the six services the reference solves there have E in {4, 2, 1, 1, 1, 1}
MEDIA_APP = {"root_op": "HTTP GET /hotels", "root": "nginx",
             "calls": {"nginx": [("user", "text", "unique-id", "movie-id")], "text": [("url-shorten", "user-mention")],
                       "user": [("compose-a",)], "unique-id": [("compose-b",)], "movie-id": [("rating",)], "rating": [("compose-c",)]}}
FANOUT_APP = {"root_op": "compose", "root": "gateway",
              "calls": {"gateway": [("auth",), ("text", "media", "user")], "text": [("url", "mention")]}}


def write_jaeger_corpus(directory, seed, n_traces, app=HOTEL_APP, concurrency=1.5, mean_service_us=800.0, gap_us=60.0,
                        sigma=0.5, t0_us=1_655_760_000_000_000):
    """Writes n_traces files <trace id>.json under `directory` and returns the list of paths.  Children lie inside
    their parent, stages follow one another, requests arrive as a Poisson process whose rate gives about
    `concurrency` requests in flight at the root."""
    import json
    import os

    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)

    def ln(mean):
        mu = np.log(mean) - 0.5 * sigma * sigma
        return max(1, int(rng.lognormal(mu, sigma)))

    services = [app["root"]]
    for callees in app["calls"].values():
        for st in callees:
            for s in st:
                if s not in services:
                    services.append(s)
    pids = {s: "p%d" % (k + 1) for k, s in enumerate(services)}
    arrivals, t, paths = [], t0_us, []
    # a first pass fixes the mean response time so that the arrival rate matches the requested concurrency
    probe = []

    def build(service, start, tid, spans, parent_sid, op):
        """server span of `service` starting at `start`; returns its end time"""
        sid = "%016x" % rng.integers(1, 2 ** 62)
        cur = start + ln(gap_us)
        for stage in app["calls"].get(service, []):
            stage_end = cur
            for callee in stage:
                c_start = cur + ln(gap_us) // 4
                c_sid = "%016x" % rng.integers(1, 2 ** 62)
                s_start = c_start + ln(gap_us) // 2
                s_end = build(callee, s_start, tid, spans, c_sid, "/%s/Handle" % callee)
                c_end = s_end + ln(gap_us) // 2
                spans.append({"traceID": tid, "spanID": c_sid, "operationName": "/%s/Handle" % callee,
                              "references": [{"refType": "CHILD_OF", "traceID": tid, "spanID": sid}], "startTime": int(c_start),
                              "duration": int(c_end - c_start), "tags": [{"key": "span.kind", "type": "string", "value": "client"}],
                              "logs": [], "processID": pids[service], "warnings": None})
                stage_end = max(stage_end, c_end)
            cur = stage_end + ln(gap_us)
        end = cur + ln(mean_service_us)
        refs = [] if parent_sid is None else [{"refType": "CHILD_OF", "traceID": tid, "spanID": parent_sid}]
        spans.append({"traceID": tid, "spanID": sid, "operationName": op, "references": refs, "startTime": int(start),
                      "duration": int(end - start), "tags": [{"key": "span.kind", "type": "string", "value": "server"}],
                      "logs": [], "processID": pids[service], "warnings": None})
        return end

    for _ in range(16):
        sp = []
        probe.append(build(app["root"], 0, "probe", sp, None, app["root_op"]))
    mean_resp = float(np.mean(probe))
    for k in range(n_traces):
        t += int(rng.exponential(mean_resp / max(concurrency, 1e-9))) + 1
        tid = "%016x" % (0x1000 + k)
        spans = []
        build(app["root"], t, tid, spans, None, app["root_op"])
        order = rng.permutation(len(spans))  # Jaeger does not store spans in any particular order
        doc = {"data": [{"traceID": tid, "spans": [spans[j] for j in order],
                         "processes": {pids[s]: {"serviceName": s, "tags": []} for s in services}, "warnings": None}],
               "total": 0, "limit": 0, "offset": 0, "errors": None}
        path = os.path.join(directory, tid + ".json")
        with open(path, "w") as f:
            json.dump(doc, f)
        paths.append(path)
    return paths


# ---------------------------------------------------------------------------------------------
# The shape alibaba-analysis/real-parser.py:308-359 writes and `--fix 5` reads (executor.py:377-448): one record per
# call under its rpc id ("0.1.2"), a server record in the callee plus -- except for the root -- a client record with the
# same timestamps in the caller; `requestType` instead of `operationName`, `caller` / `callee`, processID = service
# name, no `processes` map, millisecond timestamps x 1000.  `cart` calls itself (the reference splits such a callee off
# as a "...-loop" service) and that self-call calls on; `violations` of the traces break parent-child containment
# (the reference drops those).
ALIBABA_APP = {"root": "gw", "calls": {"gw": [("auth",), ("cart", "catalog")], "cart": [("cart",)], "cart/self": [("stock",)],
                                       "catalog": [("db",)]}}


def write_alibaba_corpus(directory, seed, n_traces, app=ALIBABA_APP, concurrency=1.5, mean_service_ms=6.0, sigma=0.5,
                         violations=0.0, t0_ms=1_655_760_000_000, project_root=None):
    """One trace per JSON file in the output format of the reference's alibaba-analysis parser.  With `project_root` the
    replica table the executor reads for --compress_factor > 1 (data/misc/service_to_replica_new.pickle, executor.py:912) is
    written too, one entry per service of the app (alibaba_replicas)."""
    import json
    import os

    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)

    def ln(mean):
        mu = np.log(mean) - 0.5 * sigma * sigma
        return max(1, int(rng.lognormal(mu, sigma)))

    def build(service, key, rpc, caller, start, tid, records, bad):
        """records of the call `rpc` handled by `service` from `start` (ms); returns its end"""
        cur = start + ln(1.0) - 1
        for k, stage in enumerate(app["calls"].get(key, [])):
            stage_end = cur
            for j, callee in enumerate(stage):
                child = "%s.%d" % (rpc, sum(len(st) for st in app["calls"][key][:k]) + j + 1)
                sub = key + "/self" if callee == service else callee
                stage_end = max(stage_end, build(callee, sub, child, service, cur, tid, records, bad))
            cur = stage_end
        end = inner_end = cur + ln(mean_service_ms)
        if bad and rpc.count(".") == 1 and rpc.endswith(".1"):
            end += 10_000                                         # recorded longer than the root span waits for it
        rec = {"traceID": tid, "startTime": int(start) * 1000, "spanID": rpc, "caller": caller, "requestType": "rpc",
               "callee": service, "interface": "if_" + service, "duration": int(end - start) * 1000,
               "tags": [{"key": "span.kind", "value": "server"}],
               "references": [] if "." not in rpc else [{"refType": "CHILD_OF", "traceID": tid, "spanID": rpc.rsplit(".", 1)[0]}],
               "processID": service}
        records.append(rec)
        if "." in rpc:
            twin = json.loads(json.dumps(rec))
            twin["tags"][0]["value"] = "client"
            twin["processID"] = caller
            records.append(twin)
        return inner_end

    probe = [build(app["root"], app["root"], "0", "USER", 0, "probe", [], False) for _ in range(16)]
    mean_resp = float(np.mean(probe))
    t, paths = t0_ms, []
    for k in range(n_traces):
        t += int(rng.exponential(mean_resp / max(concurrency, 1e-9))) + 1
        tid = "%016x" % (0x2000 + k)
        records = []
        build(app["root"], app["root"], "0", "USER", t, tid, records, rng.random() < violations)
        path = os.path.join(directory, tid + ".json")
        with open(path, "w") as f:
            json.dump({"data": [{"traceID": tid, "spans": records}]}, f)
        paths.append(path)
    if project_root is not None:
        names = sorted({app["root"]} | {c for stages in app["calls"].values() for st in stages for c in st})
        write_replica_table(project_root, dict(zip(names, alibaba_replicas(seed, len(names)))))
    return paths


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs 3-5 as unit sets.
# nodejs_microservices_with_arbitrary_file_io: 4 services, E in {1, 2, 1, 1}, millisecond-granular
# timestamps, heavily interleaved requests.
NODEJS_SERVICES = ["single", "par2", "single", "single"]

# Alibaba-shape call graphs (15 graphs, depth 2-5, fan-out 1-6, chains and parallel
# stages mixed, millisecond timestamps, no network gap): the per-service shapes that occur in them.
ALIBABA_GRAPHS = [
    ["single", "chain2"], ["par2", "single", "single"], ["chain3", "single"], ["fan6", "single", "chain2"],
    ["diamond", "par2"], ["mix7", "single"], ["chain5", "single", "single"], ["par4", "chain2", "single"],
    ["mix8", "par2"], ["chain2", "chain2", "single"], ["par2", "par2", "single"], ["diamond", "chain3"],
    ["fan6", "par4"], ["chain3", "par2", "single"], ["single", "single", "par4", "chain2"],
]


# exps/exp5/run_experiment.sh:60-156: every call graph is run at six --compress_factor values; the load factor of a service is
# max(1, ceil(compress_factor / #replicas)) (executor.py:1089-1097, data/misc/service_to_replica_new.pickle)
EXP5_COMPRESS_FACTORS = (1, 200, 1000, 4000, 10000, 15000)


def alibaba_replicas(seed, n_services, max_factor=6):
    """Replica counts for the services of a generated Alibaba-shape corpus (the table of the real corpus is not shipped):
    between 15000 / max_factor and 15000, so that exp5's six compress factors scale a service's load by 1 ... max_factor."""
    rng = np.random.default_rng(seed)
    return [int(x) for x in rng.integers(int(np.ceil(15000 / max_factor)), 15001, size=n_services)]


def write_replica_table(project_root, replicas):
    """data/misc/service_to_replica_new.pickle = {service: [replica ids]} (executor.py:912) under `project_root`."""
    import os
    import pickle

    path = os.path.join(project_root, "data", "misc", "service_to_replica_new.pickle")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump({name: list(range(int(n))) for name, n in replicas.items()}, f)
    return path


def make_nodejs_workload(seed, n_in_per_unit, concurrency=4.0, replicas=1):
    return make_workload(seed, n_in_per_unit, services=NODEJS_SERVICES, replicas=replicas, concurrency=concurrency,
                         granularity_us=1000, mean_service_us=6000.0, gap_us=1500.0)


def make_alibaba_workload(seed, total_spans, graphs=None, concurrency=1.3):
    """About `total_spans` engine spans (incoming + outgoing) spread evenly over the services of the call graphs;
    millisecond-granular, zero-gap timestamps as real-parser.py:323-353 emits them.  Returns (units, truth, graph id
    of every unit)."""
    graphs = list(range(len(ALIBABA_GRAPHS))) if graphs is None else list(graphs)
    shapes = [(g, s) for g in graphs for s in ALIBABA_GRAPHS[g]]
    per_request = sum(1 + len({e for st in SHAPES[s] for e in st}) for _, s in shapes)
    n_in = max(2, int(total_spans // per_request))
    n_in += 1 if n_in % 100 == 1 else 0   # a last parameter block of one request has no variance (hazard H3)
    units, truth, gid = [], [], []
    for k, (g, s) in enumerate(shapes):
        u, tp = make_unit(seed * 7919 + k, n_in, shape=s, concurrency=concurrency, granularity_us=1000,
                          mean_service_us=8000.0, gap_us=1.0)
        units.append(u)
        truth.append(tp)
        gid.append(g)
    return units, truth, gid
