#!/usr/bin/env python3
"""Golden vectors for the *heavy-load* regime (size-capped windows, span consumption across windows):
the reference's TraceWeaverV3.FindAssignments run directly on Span objects built from synthetic units
(traceweaver_amd/synth.py), because the shipped corpora never leave the light regime.

TEST INFRASTRUCTURE ONLY; needs /root/reference; run by hand, outputs tests/golden/refsyn_*.npz.
Same stand-ins and recorder as gen_golden.py (which see).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import gen_golden as G  # noqa: E402

# (name, seed, n_in, shape, concurrency, granularity_us)
CASES = [
    ("chain3_c6", 41, 260, "chain3", 6, 1),
    ("par2_c5", 42, 260, "par2", 5, 1),
    ("single_c10", 43, 330, "single", 10, 1),
    ("chain2_c8_ms", 44, 260, "chain2", 8, 1000),
    ("diamond_c3", 45, 240, "diamond", 3, 1),
    # 1000-request units at 6-8 requests in flight: size-capped windows with span consumption at the scale of a shipped corpus
    # (the reference takes 10 s to a few minutes each)
    ("chain3_c6_1k", 51, 1000, "chain3", 6, 1),
    ("par2_c7_1k", 52, 1000, "par2", 7, 1),
    ("single_c8_1k", 53, 1000, "single", 8, 1),
    ("chain2_c6_ms_1k", 54, 1000, "chain2", 6, 1000),
    ("chain3_c9_1k", 55, 1000, "chain3", 9, 1),
    ("par4_c4_1k", 56, 1000, "par4", 4, 1),
    ("single_c14_ms_1k", 57, 1000, "single", 14, 1000),
]


def run_case(name, seed, n, shape, conc, gran):
    import networkx as nx

    from traceweaver_amd import synth

    pydir = os.path.join(G.REF, "src", "trace_reconstructor", "ports", "python")
    saved_path, saved_mods = list(sys.path), set(sys.modules)
    try:
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = G.load_patched_v3(pydir)
        from spans import Span

        rec = G.Recorder()
        rec.install(v3mod)
        u, tp = synth.make_unit(seed, n, shape=shape, concurrency=conc, granularity_us=gran)
        E = u.E
        eps = ["ep%d" % e for e in range(E)]
        all_spans, all_processes = {}, {}
        in_spans, out_parts = [], {ep: [None] * n for ep in eps}
        for i in range(n):
            tid = "t%06d" % i
            all_processes[tid] = {"p_self": "svc"}
            root = Span(tid, "in", int(u.in_start[i]), int(u.in_end[i] - u.in_start[i]), "req", [], "p_self", "server", [])
            all_spans[root.GetId()] = root
            in_spans.append(root)
            for e, ep in enumerate(eps):
                j = int(tp[e, i])
                a = int(u.out_off[e]) + j
                c = Span(tid, "c%d" % e, int(u.out_start[a]), int(u.out_end[a] - u.out_start[a]), "call", [(tid, "in")], "p_self", "client", [])
                srv = Span(tid, "s%d" % e, int(u.out_start[a]), int(u.out_end[a] - u.out_start[a]), "call", [c.GetId()], "p_" + ep, "server", [])
                c.children_spans = [srv.GetId()]
                all_processes[tid]["p_" + ep] = ep
                all_spans[c.GetId()] = c
                all_spans[srv.GetId()] = srv
                out_parts[ep][j] = c
        graph = nx.DiGraph()
        for ep in eps:
            graph.add_node(ep)
        for p in range(E):
            for q in range(E):
                if u.dag[p, q]:
                    graph.add_edge(eps[p], eps[q])
        truth = {ep: {s.GetId(): out_parts[ep][int(tp[e, i])].GetId() for i, s in enumerate(in_spans)} for e, ep in enumerate(eps)}
        pred = v3mod.TraceWeaverV3(all_spans, all_processes)
        np.random.seed(G.SEED)
        devnull, saved = open(os.devnull, "w"), sys.stdout
        sys.stdout = devnull
        try:
            pred.FindAssignments("MaxScoreBatchSubsetWithSkips", "svc", {"client_req": in_spans}, out_parts, False, [], truth, graph)
        finally:
            sys.stdout = saved
        c = rec.services[0]
        d = G.pack_service("synthetic:" + name, c)
        d["synth"] = np.array([seed, n, conc, gran])
        d["synth_shape"] = np.array(shape)
        out = os.path.join(G.GOLDEN_DIR, "refsyn_%s.npz" % name)
        np.savez_compressed(out, **d)
        wl = c["windows"][:, 1] - c["windows"][:, 0] + 1
        aff = sum(1 for a, b in zip(c["passes"][0]["topk"], c["passes"][0]["topk2"]) if a != b)
        print("wrote", out, "wall %.1fs" % c["wall_s"], "max window", wl.max(), "spans whose top_k differs from top_k_2:", aff, flush=True)
    finally:
        sys.path[:] = saved_path
        for m in set(sys.modules) - saved_mods:
            del sys.modules[m]


if __name__ == "__main__":
    only = sys.argv[1:]
    for case in CASES:
        if only and case[0] not in only:
            continue
        run_case(*case)
