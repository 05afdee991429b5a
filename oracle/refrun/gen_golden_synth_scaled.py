#!/usr/bin/env python3
"""Golden vectors for *load-scaled* (float-timestamp) units of every DAG shape: the reference's
TraceWeaverV3.FindAssignments run directly on Span objects whose start_mus are the floats the executor's load scaling
produces (helpers/transforms.py:10-40; here via traceweaver_amd.transforms.compress_unit, which tests/test_transforms.py
shows to be identical to it on the shipped corpora).

Why next to gen_golden_compress.py: through the executor the reference's refit subtracts transformed from untransformed
timestamps (all_spans holds the originals) and usually stops after pass 1.  Here `all_spans` holds the very objects
the predictor is handed, so both passes run the way the algorithm means them to: pass 2, the gap samples and the final
refit on float timestamps are pinned as well, for chains, diamonds, parallel stages and the 7- and 8-endpoint mixes.

TEST INFRASTRUCTURE ONLY; needs /root/reference; run by hand, outputs tests/golden/refsynx_*.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import gen_golden as G  # noqa: E402

# (name, seed, n_in, shape, concurrency, granularity_us, load factor)
CASES = [
    ("chain3_x3", 51, 300, "chain3", 1.4, 1, 3),
    ("diamond_x2", 52, 260, "diamond", 1.6, 1, 2),
    ("par4_x3", 53, 240, "par4", 1.2, 1, 3),
    ("mix7_x7", 54, 200, "mix7", 0.3, 1, 7),
    ("mix8_x2", 55, 200, "mix8", 0.8, 1, 2),
    ("chain2_ms_x3", 56, 300, "chain2", 1.5, 1000, 3),
    ("single_x1", 57, 300, "single", 6.0, 1, 1),
]


def run_case(name, seed, n, shape, conc, gran, factor):
    import networkx as nx

    from traceweaver_amd import synth, transforms

    pydir = os.path.join(G.REF, "src", "trace_reconstructor", "ports", "python")
    saved_path, saved_mods = list(sys.path), set(sys.modules)
    try:
        sys.path[:0] = [os.path.join(HERE, "shims"), pydir]
        v3mod = G.load_patched_v3(pydir)
        from spans import Span

        rec = G.Recorder()
        rec.install(v3mod)
        u0, tp0 = synth.make_unit(seed, n, shape=shape, concurrency=conc, granularity_us=gran)
        s = transforms.compress_unit(u0, tp0, factor)
        u, tp = s.arrays, s.true_parent
        f_in, f_out = s.in_start, s.out_start                        # float64 starts
        in_dur = (u0.in_end - u0.in_start)[s.in_perm]
        E = u.E
        eps = ["ep%d" % e for e in range(E)]
        all_spans, all_processes = {}, {}
        in_spans, out_parts = [], {ep: [None] * n for ep in eps}
        out_dur = [(u0.out_end - u0.out_start)[int(u0.out_off[e]):int(u0.out_off[e + 1])][s.out_perm[e]] for e in range(E)]
        for i in range(n):
            tid = "t%06d" % i
            all_processes[tid] = {"p_self": "svc"}
            root = Span(tid, "in", float(f_in[i]), int(in_dur[i]), "req", [], "p_self", "server", [])
            all_spans[root.GetId()] = root
            in_spans.append(root)
            for e, ep in enumerate(eps):
                j = int(tp[e, i])
                a = int(u.out_off[e]) + j
                c = Span(tid, "c%d" % e, float(f_out[a]), int(out_dur[e][j]), "call", [(tid, "in")], "p_self", "client", [])
                srv = Span(tid, "s%d" % e, float(f_out[a]), int(out_dur[e][j]), "call", [c.GetId()], "p_" + ep, "server", [])
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
        truth = {ep: {sp.GetId(): out_parts[ep][int(tp[e, i])].GetId() for i, sp in enumerate(in_spans)} for e, ep in enumerate(eps)}
        pred = v3mod.TraceWeaverV3(all_spans, all_processes)
        np.random.seed(G.SEED)
        devnull, saved = open(os.devnull, "w"), sys.stdout
        sys.stdout = devnull
        try:
            pred.FindAssignments("MaxScoreBatchSubsetWithSkips", "svc", {"client_req": in_spans}, out_parts, False, [], truth, graph)
        finally:
            sys.stdout = saved
        c = rec.services[0]
        d = G.pack_service("synthetic-scaled:" + name, c)
        d["in_start"], d["out_start"] = f_in, f_out                  # what the predictor saw (the recorder truncates to int64)
        d["in_dur"] = in_dur
        d["out_dur"] = np.concatenate(out_dur)
        d["compress_factor"] = np.array(factor)
        d["pass1_only"] = np.array(0)
        d["synth"] = np.array([seed, n, conc, gran])
        d["synth_shape"] = np.array(shape)
        out = os.path.join(G.GOLDEN_DIR, "refsynx_%s.npz" % name)
        np.savez_compressed(out, **d)
        wl = c["windows"][:, 1] - c["windows"][:, 0] + 1
        print("wrote", out, "E", E, "wall %.1fs" % c["wall_s"], "max window", wl.max(),
              "acc pass1 %.3f final %.3f" % ((c["pass1_assign"] == c["true_parent"]).all(0).mean(), (c["final_parent"] == c["true_parent"]).all(0).mean()), flush=True)
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
