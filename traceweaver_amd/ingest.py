"""Jaeger-JSON corpora -> service units, through the native loader in libtwgpu.so (csrc/tw_ingest.cpp).

Mirrors what the reference's executor does between reading a trace directory and calling the predictor
(executor.py:287-339,755-849,1080-1135) for corpora that need no span rewriting; see include/traceweaver_amd.h.
"""
import ctypes
import os

import numpy as np

from . import _ffi
from .engine import UnitArrays


# executor.py:109-115 (process_map_1): which service calls which in the reference's nodejs application
NODEJS_CALLERS = {"service5": "service3", "service4": "service2", "service2": "service1", "service3": "service1",
                  "service1": "init-service"}

# executor.py:757-763: --fix value -> (operation name of the roots that are kept, span surgery)
REFERENCE_FIX = {0: ("init-span", "client_twins"), 1: ("ComposeReview", "reroot"), 2: ("HTTP GET /hotels", None),
                 3: ("HTTP GET /recommendations", None), 4: ("[Todo] CompleteTodoCommandHandler", None), 5: (None, "rpc_twins")}


class IngestedUnit(object):
    """One service as the hot path sees it, plus what is needed to translate results back."""

    def __init__(self, arrays, true_parent, in_trace, service, in_ep, out_eps, in_rows, out_rows, process_id=0):
        self.arrays = arrays              # UnitArrays (endpoints in topological order)
        self.true_parent = true_parent    # [E, n_in] int32 index of the true outgoing span per endpoint
        self.in_trace = in_trace          # [n_in] trace number of every incoming span
        self.service = service            # service name
        self.in_ep = in_ep                # caller ("client_<operation>" for entry services)
        self.out_eps = out_eps            # callee names, topological order
        self.in_rows = in_rows            # span-table row of every incoming span
        self.out_rows = out_rows          # per endpoint: span-table rows of its outgoing spans
        self.process_id = process_id      # position among the services that make calls (executor.py:1079)


class Corpus(object):
    def __init__(self, lib_path=None):
        self._lib = _ffi.load(lib_path)
        self._h = ctypes.c_void_p(0)
        if self._lib.tw_corpus_create(ctypes.byref(self._h)) != 0:
            raise RuntimeError("tw_corpus_create failed")

    def close(self):
        if self._h:
            self._lib.tw_corpus_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_files(self, paths, first_span=None, max_traces=1001, threads=0, fix=None, callers=None):
        """executor.py:287-339,864-874: traces in time order, roots filtered by operation name, at most
        `max_traces` kept (the reference's literal 1001; 0 = no limit).  `fix`: None / "client_twins" (FixSpans,
        nodejs corpora; `callers` = service -> calling service, default the reference's table) / "reroot" (FixSpans2,
        media corpora)."""
        mode = {None: 0, "none": 0, "client_twins": 1, "reroot": 2, "rpc_twins": 3}[fix]
        if mode == 1:
            callers = NODEJS_CALLERS if callers is None else callers
            k = (ctypes.c_char_p * len(callers))(*[x.encode() for x in callers.keys()])
            v = (ctypes.c_char_p * len(callers))(*[x.encode() for x in callers.values()])
            if self._lib.tw_corpus_set_callers(self._h, k, v, len(callers)) != 0:
                raise RuntimeError("tw_corpus_set_callers failed")
        # the path array without a Python object per path: one NUL-separated text, pointers computed from the separators
        n = len(paths)
        if all(isinstance(p, str) for p in paths):
            blob = os.fsencode("\0".join(paths)) + b"\0"
        else:
            blob = b"\0".join(os.fsencode(p) for p in paths) + b"\0"
        ends = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 0)
        if n == 0 or len(ends) != n:
            if n:
                raise ValueError("a path holds a NUL character")
            ends = np.zeros(0, dtype=np.int64)
        base = ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p).value or 0
        addr = np.empty(max(n, 1), dtype=np.uint64)
        addr[:n] = base + np.concatenate(([0], ends[:-1] + 1)).astype(np.uint64) if n else 0
        arr = ctypes.cast(ctypes.c_void_p(addr.ctypes.data), ctypes.POINTER(ctypes.c_char_p))
        rc = self._lib.tw_corpus_add_files(self._h, arr, n, first_span.encode() if first_span else None, int(max_traces), int(threads), mode)
        if rc != 0:
            raise RuntimeError("tw_corpus_add_files failed: %d" % rc)
        return self.counts()

    def add_directory(self, directory, **kw):
        files = sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.endswith("json"))
        return self.add_files(files, **kw)

    def counts(self):
        c = np.zeros(6, dtype=np.int64)
        self._lib.tw_corpus_counts(self._h, ctypes.c_void_p(c.ctypes.data))
        return dict(zip(("spans", "traces", "files", "files_rejected", "traces_filtered", "strings"), (int(x) for x in c)))

    def first_error(self):
        return self._lib.tw_corpus_last_error(self._h).decode()

    def string(self, idx):
        s = self._lib.tw_corpus_string(self._h, int(idx))
        return s.decode() if s is not None else None

    def loop_origin(self, service):
        """The service a "...-loop" stand-in was split from by the --fix 5 rewrite (executor.py:386-399), or None."""
        s = self._lib.tw_corpus_loop_origin(self._h, service.encode())
        return s.decode() if s is not None else None

    def trace_names(self):
        """String id of the traceID of every trace held (trace numbers index this array)."""
        out = np.zeros(self.counts()["traces"], dtype=np.int32)
        self._lib.tw_corpus_trace_names(self._h, ctypes.c_void_p(out.ctypes.data))
        return out

    def span_table(self):
        """Columns of the span table as numpy arrays (names as string ids, see string())."""
        n = self.counts()["spans"]
        cols = {"trace": np.empty(n, np.int32), "span_id": np.empty(n, np.int32), "service": np.empty(n, np.int32),
                "op_name": np.empty(n, np.int32), "parent": np.empty(n, np.int32), "start": np.empty(n, np.int64),
                "duration": np.empty(n, np.int64), "kind": np.empty(n, np.uint8)}
        t = _ffi.SpanTable(*[ctypes.c_void_p(cols[k].ctypes.data) for k in ("trace", "span_id", "service", "op_name", "parent", "start", "duration", "kind")])
        self._lib.tw_corpus_span_table(self._h, ctypes.byref(t))
        return cols

    def units(self):
        """Every service with a single caller and one call per (request, endpoint) -> IngestedUnit; also returns
        the counts of services left out ({"several_callers", "skip_mode", "too_small", "cyclic_order"})."""
        us = _ffi.UnitSet()
        if self._lib.tw_corpus_build_units(self._h, ctypes.byref(us)) != 0:
            raise RuntimeError("tw_corpus_build_units failed")
        n = us.n_units

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,)).copy()

        in_off = view(us.unit_in_off, n + 1, np.int64)
        E = view(us.unit_E, n, np.int32)
        n_ep = int(E.sum())
        ep_off = view(us.ep_off, n_ep + 1, np.int64)
        n_in, n_out = int(in_off[-1]) if n else 0, int(ep_off[-1]) if n else 0
        dag = view(us.dag, int((E.astype(np.int64) ** 2).sum()), np.uint8)
        key_rank = view(us.key_rank, n_ep, np.int32)
        in_start, in_end = view(us.in_start, n_in, np.int64), view(us.in_end, n_in, np.int64)
        out_start, out_end = view(us.out_start, n_out, np.int64), view(us.out_end, n_out, np.int64)
        truth = view(us.true_child, int((E.astype(np.int64) * np.diff(in_off)).sum()), np.int32)
        in_trace, in_row, out_row = view(us.in_trace, n_in, np.int32), view(us.in_row, n_in, np.int32), view(us.out_row, n_out, np.int32)
        svc, epn, inep = view(us.unit_service, n, np.int32), view(us.ep_name, n_ep, np.int32), view(us.in_ep_name, n, np.int32)
        order = view(us.unit_order, n, np.int32)
        out, ep0, d0, t0 = [], 0, 0, 0
        for u in range(n):
            e, a, b = int(E[u]), int(in_off[u]), int(in_off[u + 1])
            o0, o1 = int(ep_off[ep0]), int(ep_off[ep0 + e])
            arrays = UnitArrays(in_start[a:b], in_end[a:b], ep_off[ep0:ep0 + e + 1] - o0, out_start[o0:o1], out_end[o0:o1],
                                dag[d0:d0 + e * e].reshape(e, e), key_rank[ep0:ep0 + e])
            tp = truth[t0:t0 + e * (b - a)].reshape(e, b - a)
            rows = [out_row[int(ep_off[ep0 + k]):int(ep_off[ep0 + k + 1])] for k in range(e)]
            out.append(IngestedUnit(arrays, tp, in_trace[a:b], self.string(svc[u]), self.string(inep[u]),
                                    [self.string(x) for x in epn[ep0:ep0 + e]], in_row[a:b], rows, int(order[u])))
            ep0 += e
            d0 += e * e
            t0 += e * (b - a)
        skipped = dict(zip(("several_callers", "skip_mode", "too_small", "cyclic_order"), (int(x) for x in us.skipped)))
        return out, skipped, int(us.n_traces)
