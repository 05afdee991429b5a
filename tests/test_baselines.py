"""Baseline predictors (traceweaver_amd/baselines.py) against the reference's own classes, imported from
/root/reference (CPU tier, this container only): same inputs -> identical assignments."""
import os
import sys

import numpy as np
import pytest

from traceweaver_amd import baselines, synth
from traceweaver_amd.ingest import Corpus, REFERENCE_FIX

REF_PY = "/root/reference/src/trace_reconstructor/ports/python"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_PY), reason="the reference tree is not present")


class Span(object):
    """The attributes the baseline classes read (reference spans.py:1-75)."""

    def __init__(self, trace_id, sid, start, dur, kind):
        self.trace_id, self.sid, self.start_mus, self.duration_mus, self.span_kind = trace_id, sid, int(start), int(dur), kind

    def GetId(self):
        return (self.trace_id, self.sid)


def reference_classes():
    saved = list(sys.path)
    sys.path.insert(0, REF_PY)
    try:
        from algorithms.arrival_order import ArrivalOrder
        from algorithms.fcfs import FCFS
        from algorithms.vpath import vPath
    finally:
        sys.path[:] = saved
    return FCFS, ArrivalOrder, vPath


def protocol_inputs(u, trace_of_in, trace_of_out):
    """Partitions in partition-key order, as the executor hands them to a predictor."""
    a = u.arrays
    in_spans = [Span(trace_of_in[i], "in%d" % i, a.in_start[i], a.in_end[i] - a.in_start[i], "server") for i in range(a.n_in)]
    parts, keys = {}, [int(k) for k in np.argsort(a.key_rank, kind="stable")]
    for e in keys:
        o0, o1 = int(a.out_off[e]), int(a.out_off[e + 1])
        parts[u.out_eps[e]] = [Span(trace_of_out[e][j], "o%d_%d" % (e, j), a.out_start[o0 + j], a.out_end[o0 + j] - a.out_start[o0 + j], "client")
                               for j in range(o1 - o0)]
    truth = {u.out_eps[e]: {in_spans[i].GetId(): parts[u.out_eps[e]][u.true_parent[e, i]].GetId() for i in range(a.n_in)} for e in keys}
    return {u.in_ep: in_spans}, parts, truth


def to_parent(u, asg, in_parts, parts):
    in_spans = list(in_parts.values())[0]
    out = np.full((u.arrays.E, u.arrays.n_in), -1, dtype=np.int32)
    for e, ep in enumerate(u.out_eps):
        pos = {s.GetId(): j for j, s in enumerate(parts[ep])}
        for i, s in enumerate(in_spans):
            v = asg[ep][s.GetId()]
            out[e, i] = pos[v] if v != ("NA", "NA") else -1
    return out


def check_units(units):
    FCFS, ArrivalOrder, vPath = reference_classes()
    for u in units:
        a = u.arrays
        trace_in = ["t%d" % k for k in u.in_trace]
        trace_out = [["t%d" % u.in_trace[np.flatnonzero(u.true_parent[e] == j)[0]] if (u.true_parent[e] == j).any() else "x%d_%d" % (e, j)
                      for j in range(int(a.out_off[e + 1] - a.out_off[e]))] for e in range(a.E)]
        in_parts, parts, truth = protocol_inputs(u, trace_in, trace_out)
        for name, cls, mine in (("FCFS", FCFS, baselines.fcfs(a)), ("ArrivalOrder", ArrivalOrder, baselines.arrival_order(a)),
                                ("vPath", vPath, baselines.vpath(a, u.true_parent))):
            import copy
            ref = cls({}, {}).FindAssignments(name, u.service, copy.deepcopy(in_parts), copy.deepcopy(parts), False, [], truth)
            assert np.array_equal(to_parent(u, ref, in_parts, parts), mine), (u.service, name)


def check_wap5(units):
    """One reference instance and one of ours, fed the services in the same order (state leaks between services)."""
    import copy

    saved = list(sys.path)
    sys.path.insert(0, REF_PY)
    try:
        from algorithms.wap5 import WAP5
    finally:
        sys.path[:] = saved
    ref, mine = WAP5({}, {}), baselines.WAP5()
    for u in units:
        a = u.arrays
        trace_in = ["t%d" % k for k in u.in_trace]
        trace_out = [["x%d_%d" % (e, j) for j in range(int(a.out_off[e + 1] - a.out_off[e]))] for e in range(a.E)]
        in_parts, parts, truth = protocol_inputs(u, trace_in, trace_out)
        got = ref.FindAssignments("WAP5", u.service, copy.deepcopy(in_parts), copy.deepcopy(parts), False, [], truth)
        options = mine.assign(a, u.out_eps)
        in_spans = list(in_parts.values())[0]
        for e, ep in enumerate(u.out_eps):
            pos = {s.GetId(): j for j, s in enumerate(parts[ep])}
            for i, s in enumerate(in_spans):
                want = [pos[x] for x in got[ep][s.GetId()] if tuple(x) != ("NA", "NA")]
                assert options[e][i] == want, (u.service, ep, i)


def test_wap5_on_generated_corpora(emu_lib, tmp_path):
    synth.write_jaeger_corpus(str(tmp_path), 23, 300, app=synth.HOTEL_APP, concurrency=2.0)
    c = Corpus(lib_path=emu_lib)
    c.add_directory(str(tmp_path), first_span="HTTP GET /hotels", max_traces=0)
    check_wap5(c.units()[0])
    c.close()


@pytest.mark.parametrize("rel,fix", [("hotel_reservation/hotel_load100", 2), ("nodejs_microservices/node_load150", 0)])
def test_wap5_on_reference_corpora(emu_lib, rel, fix):
    first, surgery = REFERENCE_FIX[fix]
    c = Corpus(lib_path=emu_lib)
    c.add_directory("/root/reference/data/" + rel, first_span=first, max_traces=1001, fix=surgery)
    check_wap5(c.units()[0])
    c.close()


def test_baselines_on_generated_corpora(emu_lib, tmp_path):
    for app, conc in ((synth.HOTEL_APP, 2.5), (synth.FANOUT_APP, 1.6)):
        d = tmp_path / app["root"]
        synth.write_jaeger_corpus(str(d), 21, 250, app=app, concurrency=conc)
        c = Corpus(lib_path=emu_lib)
        c.add_directory(str(d), first_span=app["root_op"], max_traces=0)
        units, _, _ = c.units()
        check_units(units)
        c.close()


@pytest.mark.parametrize("rel,fix", [("hotel_reservation/hotel_load150", 2), ("nodejs_microservices_with_arbitrary_file_io/node_1", 0),
                                     ("media_microservices/media_load100", 1)])
def test_baselines_on_reference_corpora(emu_lib, rel, fix):
    first, surgery = REFERENCE_FIX[fix]
    c = Corpus(lib_path=emu_lib)
    c.add_directory("/root/reference/data/" + rel, first_span=first, max_traces=1001, fix=surgery)
    units, _, _ = c.units()
    check_units(units)
    c.close()
