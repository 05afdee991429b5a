"""Jaeger-JSON corpora -> service units, through the native loader in libtwgpu.so (csrc/tw_ingest.cpp).

Mirrors what the reference's executor does between reading a trace directory and calling the predictor
(executor.py:287-339,755-849,1080-1135) for corpora that need no span rewriting; see include/traceweaver_amd.h.
"""
import ctypes
import json
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
        files = sorted(os.path.join(directory, f) for f in os.listdir(directory) if _is_trace_file(f))
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

        self._last_unit_set = None
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
        self._last_unit_set = {"in_off": in_off, "E": E, "ep_off": ep_off, "dag": dag, "key_rank": key_rank, "in_start": in_start, "in_end": in_end,
                               "out_start": out_start, "out_end": out_end, "truth": truth, "in_trace": in_trace, "in_row": in_row, "out_row": out_row,
                               "order": order, "_units": out, "_skipped": skipped, "_n_traces": int(us.n_traces)}
        return out, skipped, int(us.n_traces)

    def _unit_set_arrays(self):
        """The flat arrays units() was built from (what the directory cache stores)."""
        self.units()
        return dict(self._last_unit_set)


# ---------------------------------------------------------------------------------------------------------------------
# The span table of a directory, kept next to its traces ("a one-time binary cache next to
# time_order_filenames.pickle").  The reference keeps the time-ordered file list of a directory in a pickle inside it and
# trusts it until --clear_cache 1 (executor.py:320-339); this cache holds what the native loader made of the files -- span
# table, units, the strings results are keyed by -- under the same rule, plus a cheap guard: loader version and sources,
# the arguments of the load and the directory's modification time (entries added, removed or renamed; a file rewritten in
# place is not seen -- that is what --clear_cache is for).  A hit replaces reading and parsing every JSON file.
CACHE_FILE = "tw_span_table.bin"
CACHE_VERSION = 1
_UNIT_ARRAYS = ("in_off", "E", "ep_off", "dag", "key_rank", "in_start", "in_end", "out_start", "out_end", "truth", "in_trace", "in_row", "out_row", "order")
_TABLE_COLUMNS = ("trace", "span_id", "service", "op_name", "parent", "start", "duration", "kind")


def _cache_key(directory, first_span, max_traces, fix, callers):
    from . import build

    return json.dumps({"version": CACHE_VERSION, "loader": build.source_digest(),
                       "first_span": first_span, "max_traces": int(max_traces), "fix": fix,
                       "callers": sorted((NODEJS_CALLERS if callers is None else callers).items()) if fix == "client_twins" else None}, sort_keys=True)


_CACHE_MAGIC = b"TWSPANS1"


def _write_arrays(path, header, arrays):
    """One file: magic, length of the JSON header, the header (with every array's dtype, shape and offset), then the arrays
    on 64-byte boundaries -- read back as views of one memory map, nothing is copied or checksummed."""
    directory, at = {}, 0
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        arrays[name] = a
        directory[name] = (a.dtype.str, list(a.shape), at)
        at += (a.nbytes + 63) // 64 * 64
    head = json.dumps(dict(header, arrays=directory)).encode()
    pad = (-(len(_CACHE_MAGIC) + 8 + len(head))) % 64
    # never rewritten in place: readers hand out views of a memory map of the file (several executors of one exps script
    # share a data directory) -- the new file replaces the old name, a running reader keeps the old inode
    tmp = "%s.%d.tmp" % (path, os.getpid())
    try:
        with open(tmp, "wb") as f:
            f.write(_CACHE_MAGIC + np.uint64(len(head) + pad).tobytes() + head + b" " * pad)
            for name, a in arrays.items():
                f.write(a.tobytes())
                f.write(b"\0" * ((-a.nbytes) % 64))
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _read_arrays(path):
    """Header and arrays of a cache file.  Anything that is not a complete file of this version raises ValueError (or OSError):
    the caller counts that as a miss."""
    import mmap

    with open(path, "rb") as f:
        m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    if len(m) < 16 or m[:8] != _CACHE_MAGIC:
        raise ValueError("not a span table cache")
    n = int(np.frombuffer(m, dtype=np.uint64, count=1, offset=8)[0])
    if 16 + n > len(m):
        raise ValueError("span table cache: truncated header")
    header = json.loads(m[16:16 + n].decode())   # (json.JSONDecodeError and UnicodeDecodeError are ValueErrors)
    if not isinstance(header, dict) or not isinstance(header.get("arrays"), dict):
        raise ValueError("span table cache: header of another version")
    base = 16 + n
    arrays = {}
    for name, entry in header.pop("arrays").items():
        try:
            dtype, shape, off = entry
            dtype, shape, off = np.dtype(dtype), tuple(int(x) for x in shape), int(off)
        except (TypeError, ValueError) as exc:
            raise ValueError("span table cache: malformed array entry %r" % (name,)) from exc
        count = int(np.prod(shape)) if shape else 1
        if off < 0 or base + off + count * dtype.itemsize > len(m):
            raise ValueError("span table cache: array %r reaches beyond the file" % (name,))
        arrays[name] = np.frombuffer(m, dtype=dtype, count=count, offset=base + off).reshape(shape)
    return header, arrays


class CachedCorpus(object):
    """What Corpus offers after add_directory(), read back from the directory's cache file (read-only views of it)."""

    def __init__(self, meta, z):
        self._counts, self._skipped, self._n_traces = meta["counts"], meta["skipped"], meta["n_traces"]
        self._unit_names, self._loop_origin = meta["unit_names"], meta["loop_origin"]
        self._units = {k: z["u_" + k] for k in _UNIT_ARRAYS}
        self._table = {k: z["t_" + k] for k in _TABLE_COLUMNS}
        self._trace_names = z["trace_names"]
        self._handles, self._str_off, self._str_blob = z["str_handle"], z["str_off"], z["str_blob"]
        self.from_cache = True

    def close(self):
        pass

    def counts(self):
        return dict(self._counts)

    def first_error(self):
        return ""

    def string(self, idx):
        k = int(np.searchsorted(self._handles, idx))
        if k >= len(self._handles) or self._handles[k] != idx:
            return None
        return self._str_blob[int(self._str_off[k]):int(self._str_off[k + 1])].tobytes().decode()

    def loop_origin(self, service):
        return self._loop_origin.get(service)

    def trace_names(self):
        return self._trace_names

    def span_table(self):
        return dict(self._table)

    def units(self):
        a = self._units
        in_off, E, ep_off = a["in_off"], a["E"], a["ep_off"]
        out, ep0, d0, t0 = [], 0, 0, 0
        for u in range(len(E)):
            e, lo, hi = int(E[u]), int(in_off[u]), int(in_off[u + 1])
            o0, o1 = int(ep_off[ep0]), int(ep_off[ep0 + e])
            arrays = UnitArrays(a["in_start"][lo:hi], a["in_end"][lo:hi], ep_off[ep0:ep0 + e + 1] - o0, a["out_start"][o0:o1], a["out_end"][o0:o1],
                                a["dag"][d0:d0 + e * e].reshape(e, e), a["key_rank"][ep0:ep0 + e])
            rows = [a["out_row"][int(ep_off[ep0 + k]):int(ep_off[ep0 + k + 1])] for k in range(e)]
            service, in_ep, out_eps = self._unit_names[u]
            out.append(IngestedUnit(arrays, a["truth"][t0:t0 + e * (hi - lo)].reshape(e, hi - lo), a["in_trace"][lo:hi], service, in_ep, list(out_eps),
                                    a["in_row"][lo:hi], rows, int(a["order"][u])))
            ep0 += e
            d0 += e * e
            t0 += e * (hi - lo)
        return out, dict(self._skipped), int(self._n_traces)


def _is_trace_file(name):
    """What the loader reads from a directory (the reference's filter, executor.py:331): one predicate for the loader and the cache."""
    return name.endswith("json")


def _dir_state(directory):
    """(modification time, number of trace files) of the directory: what must not move while it is being loaded."""
    with os.scandir(directory) as it:
        n = sum(1 for e in it if _is_trace_file(e.name))
    return os.stat(directory).st_mtime_ns, n


def _cache_fresh(directory, path):
    """The cache file carries the directory's modification time as it was right after the file was put in place (stamped by
    _write_arrays' caller only if no trace file came or went while the directory was being loaded): entries added to, removed from
    or renamed in the directory since then have moved it on.  One stat each -- a hit must not cost a listing of 20 000 files.  Like
    the reference's cache of a directory (its time-ordered file list, executor.py:320-339) this does not see a trace file edited in
    place, and a file system with coarse time stamps can miss a change within one tick: that is what --clear_cache is for."""
    return os.stat(directory).st_mtime_ns == os.stat(path).st_mtime_ns


def _save_cache(corpus, directory, key_args, counts, state_before):
    """Writes the cache of a freshly loaded directory (to a temporary name, then renamed over the cache file) and stamps it with the
    directory's modification time after the rename -- unless the directory's (modification time, number of trace files) just before
    the cache file is written differs from `state_before`, taken before the load: then an entry arrived, left or was renamed while
    the directory was being read (a rename keeps the count and moves the time), and the cache stays unstamped = stale."""
    unchanged = _dir_state(directory) == state_before
    path = os.path.join(directory, CACHE_FILE)
    key = _cache_key(directory, *key_args)
    raw = corpus._unit_set_arrays()
    table = corpus.span_table()
    names = corpus.trace_names()
    units, skipped, n_traces = raw.pop("_units"), raw.pop("_skipped"), raw.pop("_n_traces")
    handles = np.unique(np.concatenate([table["span_id"], table["service"], table["op_name"], names]).astype(np.int64))
    strings = [corpus.string(h) for h in handles]
    keep = np.array([x is not None for x in strings], dtype=bool)
    handles, strings = handles[keep], [x.encode() for x in strings if x is not None]
    off = np.zeros(len(strings) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in strings], out=off[1:])
    services = sorted({u.service for u in units} | {corpus.string(h) for h in np.unique(table["service"]) if corpus.string(h) is not None})
    loops = {sv: corpus.loop_origin(sv) for sv in services}
    header = {"key": key, "counts": counts, "skipped": skipped, "n_traces": n_traces,
              "unit_names": [(u.service, u.in_ep, list(u.out_eps)) for u in units], "loop_origin": {k: v for k, v in loops.items() if v is not None}}
    arrays = {"trace_names": names, "str_handle": handles, "str_off": off, "str_blob": np.frombuffer(b"".join(strings), dtype=np.uint8)}
    arrays.update({"u_" + k: raw[k] for k in _UNIT_ARRAYS})
    arrays.update({"t_" + k: table[k] for k in _TABLE_COLUMNS})
    _write_arrays(path, header, arrays)
    m = os.stat(directory).st_mtime_ns   # the directory's modification time after the rename: the file carries it as its own (_cache_fresh)
    if not unchanged or _dir_state(directory)[1] != state_before[1]:
        m = 0
    os.utime(path, ns=(m, m))


def open_directory(directory, lib_path=None, first_span=None, max_traces=1001, fix=None, callers=None, threads=0, cache=True, clear_cache=False):
    """The traces of a directory as a corpus: from its cache file when there is a valid one (and `cache` / not `clear_cache`),
    otherwise through the native loader -- which then leaves the cache behind (silently not, in a read-only directory).
    Returns (Corpus or CachedCorpus, counts)."""
    path = os.path.join(directory, CACHE_FILE)
    key_args = (first_span, max_traces, fix, callers)
    if clear_cache and os.path.exists(path):
        try:
            os.remove(path)
        except OSError:
            cache = False
    if cache and os.path.exists(path):
        try:
            meta, z = _read_arrays(path) if _cache_fresh(directory, path) else ({}, None)
            if meta.get("key") == _cache_key(directory, *key_args):
                c = CachedCorpus(meta, z)
                return c, c.counts()
        except (OSError, ValueError, KeyError):   # unreadable / incomplete / written by another version (_read_arrays raises ValueError for
            pass                                  # every malformed header; CachedCorpus KeyError for a missing array): a miss
    state_before = _dir_state(directory) if cache else (0, 0)
    corpus = Corpus(lib_path=lib_path)
    counts = corpus.add_directory(directory, first_span=first_span, max_traces=max_traces, fix=fix, callers=callers, threads=threads)
    corpus.from_cache = False
    if cache:
        try:
            _save_cache(corpus, directory, key_args, counts, state_before)
        except OSError:   # read-only directory, disk full, ...: a problem with the cache FILE never fails the run (anything else is a bug and shows)
            pass
    return corpus, counts
