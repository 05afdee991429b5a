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
        # after the executor's load scaling start_mus is a float (helpers/transforms.py:21,30); durations stay ints
        self.trace_id, self.sid, self.duration_mus, self.span_kind = trace_id, sid, int(dur), kind
        self.start_mus = float(start) if isinstance(start, float) else int(start)

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
    if a.time_scale is None:
        i_s, i_d, o_s, o_d = a.in_start, a.in_end - a.in_start, a.out_start, a.out_end - a.out_start
    else:   # a load-scaled unit: the float starts it is the exact image of, the integer durations
        i_s, o_s = a.in_start * a.time_scale, a.out_start * a.time_scale
        i_d, o_d = np.rint((a.in_end - a.in_start) * a.time_scale), np.rint((a.out_end - a.out_start) * a.time_scale)
        assert np.array_equal(i_s + i_d, a.in_end * a.time_scale) and np.array_equal(o_s + o_d, a.out_end * a.time_scale)
    in_spans = [Span(trace_of_in[i], "in%d" % i, i_s[i], i_d[i], "server") for i in range(a.n_in)]
    parts, keys = {}, [int(k) for k in np.argsort(a.key_rank, kind="stable")]
    for e in keys:
        o0, o1 = int(a.out_off[e]), int(a.out_off[e + 1])
        parts[u.out_eps[e]] = [Span(trace_of_out[e][j], "o%d_%d" % (e, j), o_s[o0 + j], o_d[o0 + j], "client")
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


def _scaled(units, factor):
    """The units after the executor's load scaling (traceweaver_amd/transforms.py)."""
    from traceweaver_amd import transforms
    from traceweaver_amd.ingest import IngestedUnit

    out = []
    for u in units:
        s = transforms.compress_unit(u.arrays, u.true_parent, factor, trace_key=["t%d" % k for k in u.in_trace])
        out.append(IngestedUnit(s.arrays, s.true_parent, u.in_trace[s.in_perm], u.service, u.in_ep, u.out_eps, u.in_rows[s.in_perm],
                                [r[p] for r, p in zip(u.out_rows, s.out_perm)], u.process_id))
    return out


@pytest.mark.parametrize("factor", [3, 7])
def test_baselines_on_load_scaled_units(emu_lib, tmp_path, factor):
    """--compress_factor > 1 hands the reference's classes float timestamps; ours read the exact integer image of them."""
    synth.write_jaeger_corpus(str(tmp_path), 29, 300, app=synth.HOTEL_APP, concurrency=1.2)
    c = Corpus(lib_path=emu_lib)
    c.add_directory(str(tmp_path), first_span="HTTP GET /hotels", max_traces=0)
    units = _scaled(c.units()[0], factor)
    assert all(u.arrays.time_scale < 1.0 for u in units)
    check_units(units)
    check_wap5(units)
    c.close()
