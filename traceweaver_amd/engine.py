"""numpy-level wrapper around the C-ABI (include/traceweaver_amd.h).

A *unit* is one call of the reference's TraceWeaverV3.FindAssignments (traceweaver_v3.py:1087): one
service, its incoming spans and its outgoing spans per endpoint.  Units of a batch are independent and
are solved together on one GPU.
"""
import ctypes
import os
import warnings

import numpy as np

from . import _ffi


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (_ffi.STATUS.get(code, "TW_ERR"), code, message))
        self.code = code


class UnitArrays(object):
    """Structure-of-arrays form of one unit: endpoints in topological order of the call-order DAG
    (traceweaver_v1.py:37-39), every span list sorted by (start, end) (executor.py:1112)."""

    AFTER_CUT, BEFORE_CUT = 1, 2   # tw_batch.unit_part

    def __init__(self, in_start, in_end, out_off, out_start, out_end, dag, key_rank=None, time_scale=None, part=0):
        # time_scale: None = int64 microseconds.  A power of two = the timestamps are exact images of binary64
        # values in units of time_scale microseconds (load-scaled units, traceweaver_amd.transforms).
        self.time_scale = None if time_scale is None else float(time_scale)
        # part: the unit is a stretch of a service's requests between cuts (sharding.split_unit): AFTER_CUT / BEFORE_CUT
        self.part = int(part)
        self.in_start = np.ascontiguousarray(in_start, dtype=np.int64)
        self.in_end = np.ascontiguousarray(in_end, dtype=np.int64)
        self.out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        self.out_start = np.ascontiguousarray(out_start, dtype=np.int64)
        self.out_end = np.ascontiguousarray(out_end, dtype=np.int64)
        self.E = len(self.out_off) - 1
        self.n_in = len(self.in_start)
        self.dag = np.ascontiguousarray(dag, dtype=np.uint8).reshape(self.E, self.E)
        self.key_rank = np.ascontiguousarray(np.arange(self.E) if key_rank is None else key_rank, dtype=np.int32)
        if len(self.in_end) != self.n_in or len(self.out_start) != self.out_off[-1] or len(self.out_end) != len(self.out_start):
            raise ValueError("inconsistent unit arrays")

    @property
    def nslot(self):
        return self.E * self.E + 2 * self.E

    @property
    def n_spans(self):
        return self.n_in + len(self.out_start)


def _vp(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


class _PinnedPool(object):
    """Page-locked staging arrays for Engine.load (tw_host_alloc): the concatenation of the units' arrays has to be
    written somewhere anyway -- written into pinned memory, tw_load_batch moves it at the full host-link rate."""

    def __init__(self, lib):
        self._lib = lib
        self._bufs = {}

    def array(self, name, count, dtype):
        dtype = np.dtype(dtype)
        need = max(int(count) * dtype.itemsize, 8)
        buf = self._bufs.get(name)
        if buf is None or buf[1] < need:
            if buf is not None:
                self._lib.tw_host_free(buf[0])
            p = ctypes.c_void_p(0)
            cap = need + need // 4
            if self._lib.tw_host_alloc(cap, ctypes.byref(p)) != 0 or not p.value:
                raise MemoryError("tw_host_alloc(%d) failed" % cap)
            buf = (p, cap)
            self._bufs[name] = buf
        raw = (ctypes.c_char * need).from_address(buf[0].value)
        return np.frombuffer(raw, dtype=dtype, count=int(count))

    def close(self):
        for p, _ in self._bufs.values():
            self._lib.tw_host_free(p)
        self._bufs = {}


class Engine(object):
    def __init__(self, device=0, lib_path=None):
        self._lib = _ffi.load(lib_path)
        self._h = ctypes.c_void_p(0)
        rc = self._lib.tw_create(int(device), ctypes.byref(self._h))
        if rc != 0:
            raise EngineError(rc, "tw_create failed on device %d" % device)
        self.units = []
        self._keep = None
        self._n_traces = 0
        self._pinned = _PinnedPool(self._lib)
        self.last_load_s = {}

    def close(self):
        if self._h:
            self._keep = None
            self._pinned.close()
            self._lib.tw_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self._lib.tw_last_error(self._h).decode())

    # ------------------------------------------------------------------------------------------
    _warned_queues = False

    def load(self, units, batch_size=100, batch_size_mis=30, skip=None):
        """Copy a list of UnitArrays into HBM.  skip: None, or per unit a skipmode.SkipPlan (time windows, skip-span pools,
        (mean, std) table) -- the batch then runs the reference's one-pass skip mode (run_pass1 only)."""
        self.units = list(units)
        self.skip_mode = skip is not None
        in_off = np.zeros(len(units) + 1, dtype=np.int64)
        np.cumsum([u.n_in for u in units], out=in_off[1:])
        unit_E = np.array([u.E for u in units], dtype=np.int32)
        # every endpoint-count class runs on a stream of its own; the HIP runtime maps a process' streams onto GPU_MAX_HW_QUEUES
        # hardware queues (4 unless the host exports more BEFORE HIP initialises) and classes that share a queue run one after the
        # other: 9.5 instead of 6.3 ms per pass on an eight-class batch (tw_create).  The library leaves the variable alone
        # (it holds for the whole process): tell a host that embeds the engine, once.
        if len(set(unit_E.tolist())) > 3 and "GPU_MAX_HW_QUEUES" not in os.environ and not Engine._warned_queues:
            Engine._warned_queues = True
            warnings.warn("traceweaver_amd: this batch has %d endpoint-count classes and GPU_MAX_HW_QUEUES is not set: their streams share the "
                          "runtime's 4 hardware queues and partly serialise.  Export GPU_MAX_HW_QUEUES=12 before the process touches the GPU "
                          "(or TW_SET_HW_QUEUES=1 before importing traceweaver_amd)." % len(set(unit_E.tolist())))
        ep_off = [0]
        base = 0
        for u in units:
            ep_off.extend((base + u.out_off[1:]).tolist())
            base += int(u.out_off[-1])
        arrays = {
            "unit_in_off": in_off, "unit_E": unit_E, "ep_off": np.array(ep_off, dtype=np.int64),
            "dag": np.concatenate([u.dag.ravel() for u in units]),
            "key_rank": np.concatenate([u.key_rank for u in units]),
        }
        self._keep = None   # views of the pinned buffers of the previous batch
        n_in_total, n_out_total = int(in_off[-1]), int(base)
        for name, total in (("in_start", n_in_total), ("in_end", n_in_total), ("out_start", n_out_total), ("out_end", n_out_total)):
            dst = self._pinned.array(name, total, np.int64)   # the span arrays (16 B per span) go through page-locked memory
            np.concatenate([getattr(u, name) for u in units], out=dst)
            arrays[name] = dst
        scaled = [u.time_scale is not None for u in units]
        if any(scaled) and not all(scaled):
            raise ValueError("a batch holds either integer-microsecond units or load-scaled units, not both")
        arrays["unit_time_scale"] = np.array([u.time_scale for u in units], dtype=np.float64) if all(scaled) else None
        arrays["unit_part"] = np.array([u.part for u in units], dtype=np.uint8) if any(u.part for u in units) else None
        skip_arr = None
        if skip is not None:
            if len(skip) != len(units):
                raise ValueError("one skip plan per unit")
            skip_arr = (_ffi.SkipUnit * len(units))()
            for k, (u, sp) in enumerate(zip(units, skip)):
                tw = np.ascontiguousarray(sp.tw_start, dtype=np.int64)
                pool = np.ascontiguousarray(sp.pool, dtype=np.int32).reshape(u.E, len(tw))
                dist = np.ascontiguousarray(sp.dist, dtype=np.float64).reshape(u.E + 1, u.E + 1, 2)
                arrays["skip%d" % k] = (tw, pool, dist)
                skip_arr[k] = _ffi.SkipUnit(len(tw), _vp(tw), _vp(pool), _vp(dist))
            arrays["skip"] = skip_arr
        self._keep = arrays
        b = _ffi.Batch(len(units), *[_vp(arrays[k]) for k in (
            "unit_in_off", "unit_E", "ep_off", "dag", "key_rank", "in_start", "in_end", "out_start", "out_end")],
            batch_size, batch_size_mis, _ffi.TW_TOPK, _vp(arrays["unit_time_scale"]),
            ctypes.cast(skip_arr, ctypes.c_void_p) if skip_arr is not None else ctypes.c_void_p(0), _vp(arrays["unit_part"]))
        self._check(self._lib.tw_load_batch(self._h, ctypes.byref(b), 0))
        self._in_off = in_off
        self._ie_off = np.concatenate([[0], np.cumsum([u.n_in * u.E for u in units])]).astype(np.int64)
        self._slot_off = np.concatenate([[0], np.cumsum([u.nslot for u in units])]).astype(np.int64)
        self._gap_off = np.concatenate([[0], np.cumsum([u.nslot * u.n_in for u in units])]).astype(np.int64)
        self._nblk = [(u.n_in + batch_size - 1) // batch_size for u in units]
        self._gp_off = np.concatenate([[0], np.cumsum([nb * u.nslot for nb, u in zip(self._nblk, units)])]).astype(np.int64)

    def run_pass1(self):
        self._check(self._lib.tw_run_pass1(self._h))

    def run_pass2(self):
        self._check(self._lib.tw_run_pass2(self._h))

    def gaps(self):
        """Per unit: [nslot, n_in] gap samples of the pass-1 assignment (NaN = dropped / unscored slot)."""
        g = np.empty(int(self._gap_off[-1]), dtype=np.float64)
        self._check(self._lib.tw_get_gaps(self._h, _vp(g)))
        return [g[self._gap_off[k]:self._gap_off[k + 1]].reshape(u.nslot, u.n_in) for k, u in enumerate(self.units)]

    def set_gaps(self, gaps):
        """Per unit [nslot, n_in] gap samples (NaN = none) in place of the ones pass 1 produced: the input of fit_mixtures
        for a service whose requests were solved in parts (traceweaver_amd/sharding.py)."""
        g = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in gaps]))
        if len(g) != self._gap_off[-1]:
            raise ValueError("gap rows do not match the loaded batch")
        self._check(self._lib.tw_set_gaps(self._h, _vp(g)))

    class _DeviceArray(object):
        """A device buffer of the engine as the CUDA array interface describes it (torch.as_tensor takes it without a copy)."""

        def __init__(self, ptr, count, typestr):
            self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(count),), "typestr": typestr, "version": 2}

    def device_views(self):
        """(parent, gaps): the engine's device buffers of the two exchange steps (tw_device_buffers) as objects with
        `__cuda_array_interface__` -- int32 [sum E * n_in] parent indices of the last pass run, float64 gap rows of pass 1;
        `torch.as_tensor(view, device="cuda")` wraps them in place for an all-gather over RCCL (sharding.py)."""
        v = _ffi.DeviceView()
        self._check(self._lib.tw_device_buffers(self._h, ctypes.byref(v)))
        return (Engine._DeviceArray(v.parent or 0, v.parent_count, "<i4"), Engine._DeviceArray(v.gaps or 0, v.gaps_count, "<f8"))

    def set_gaps_device(self, ptr):
        """set_gaps with a device pointer (int): the gathered rows stay in HBM."""
        self._check(self._lib.tw_set_gaps_device(self._h, ctypes.c_void_p(int(ptr))))

    def set_mixtures(self, mix_n, mix_p):
        """mix_n / mix_p: per unit arrays [nslot] int32 and [nslot, 5, 3] (weight, mean, precision_cholesky)."""
        n = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a in mix_n]), dtype=np.int32)
        p = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1, _ffi.TW_MAX_COMP, 3) for a in mix_p]))
        if len(n) != self._slot_off[-1] or len(p) != len(n):
            raise ValueError("mixture tables do not match the loaded batch")
        self._check(self._lib.tw_set_mixtures(self._h, _vp(n), _vp(p)))

    FIT_ROW_DRAWS = (0, 1, 4, 11, 21, 34)   # uniforms the model-selection fits of a row draw, by min(5, #unique)

    def fit_mixtures(self, tape=None, slot_off=None, seed=None, unit_seeds=None):
        """The reference's refit of every scored edge on the device, from the pass-1 gap samples (csrc/tw_fit.h).
        Without a tape the k-means++ draws come from the engine's own MT19937 (`seed` reseeds it first) or, with
        `unit_seeds`, from one MT19937 per unit (a unit's fit then does not depend on its batch); with (tape, slot_off) --
        per unit arrays of offsets [nslot] into one tape -- from the caller (see fit_rows)."""
        if unit_seeds is not None:
            sd = np.ascontiguousarray(unit_seeds, dtype=np.uint32)
            if len(sd) != len(self.units):
                raise ValueError("one seed per unit")
            self._check(self._lib.tw_fit_mixtures_seeded(self._h, _vp(sd)))
            return
        if tape is None:
            if seed is not None:
                self._check(self._lib.tw_set_fit_seed(self._h, int(seed)))
            self._check(self._lib.tw_fit_mixtures(self._h))
            return
        t = np.ascontiguousarray(tape, dtype=np.float64)
        o = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int64).ravel() for a in slot_off]), dtype=np.int64)
        if len(o) != self._slot_off[-1]:
            raise ValueError("tape offsets do not match the loaded batch")
        self._check(self._lib.tw_fit_mixtures_tape(self._h, _vp(t), len(t), _vp(o)))

    def fit_rows(self):
        """Per unit [nslot]: min(5, #distinct samples) of every gap row of the pass-1 result (0 = nothing to fit) -- the
        number of model-selection fits of that edge, hence the uniforms they draw (FIT_ROW_DRAWS)."""
        m = np.empty(int(self._slot_off[-1]), dtype=np.int32)
        self._check(self._lib.tw_fit_rows(self._h, _vp(m)))
        return [m[self._slot_off[k]:self._slot_off[k + 1]] for k in range(len(self.units))]

    def mixtures(self):
        """Per unit (mix_n [nslot], mix_p [nslot, 5, 3]) currently resident on the device."""
        n = np.empty(int(self._slot_off[-1]), dtype=np.int32)
        p = np.empty((int(self._slot_off[-1]), _ffi.TW_MAX_COMP, 3), dtype=np.float64)
        self._check(self._lib.tw_get_mixtures(self._h, _vp(n), _vp(p)))
        return [(n[self._slot_off[k]:self._slot_off[k + 1]], p[self._slot_off[k]:self._slot_off[k + 1]])
                for k in range(len(self.units))]

    def gauss_params(self):
        """Per unit: [nblk, nslot, 3] = mean, std, log(std used) of pass 1 (NaN for unscored slots)."""
        g = np.empty((int(self._gp_off[-1]), 3), dtype=np.float64)
        self._check(self._lib.tw_get_gauss_params(self._h, _vp(g)))
        return [g[self._gp_off[k]:self._gp_off[k + 1]].reshape(self._nblk[k], u.nslot, 3) for k, u in enumerate(self.units)]

    _FIELDS = ("parent", "topk_idx", "topk_score", "topk_n", "chosen", "leaves", "window_end", "unit_stats")

    def results(self, which, fields=None):
        """Per unit dict: parent [E,n], topk_idx [5,E,n], topk_score [5,n], topk_n, chosen, leaves, window_end
        and the counters.  `fields` restricts what is copied back from the device."""
        fields = self._FIELDS if fields is None else tuple(fields)
        n_in = int(self._in_off[-1])
        n_ie = int(self._ie_off[-1])
        K = _ffi.TW_TOPK
        shapes = {
            "parent": (n_ie, np.int32), "topk_idx": (n_ie * K, np.int32), "topk_score": (n_in * K, np.float64),
            "topk_n": (n_in, np.int32), "chosen": (n_in, np.int32), "leaves": (n_in, np.int64),
            "window_end": (n_in, np.uint8), "unit_stats": ((len(self.units), 8), np.int64),
        }
        buf = {k: (np.empty(*shapes[k]) if k in fields else None) for k in self._FIELDS}
        r = _ffi.Results(*[_vp(buf[k]) for k in self._FIELDS])
        self._check(self._lib.tw_get_results(self._h, int(which), ctypes.byref(r)))
        out = []
        for k, u in enumerate(self.units):
            a, b = int(self._in_off[k]), int(self._in_off[k + 1])
            ia, ib = int(self._ie_off[k]), int(self._ie_off[k + 1])
            o = {}
            if buf["parent"] is not None:
                o["parent"] = buf["parent"][ia:ib].reshape(u.E, u.n_in)
            if buf["topk_idx"] is not None:
                o["topk_idx"] = buf["topk_idx"][K * ia:K * ib].reshape(K, u.E, u.n_in)
            if buf["topk_score"] is not None:
                o["topk_score"] = buf["topk_score"][K * a:K * b].reshape(K, u.n_in)
            for name in ("topk_n", "chosen", "leaves", "window_end"):
                if buf[name] is not None:
                    o[name] = buf[name][a:b]
            if buf["unit_stats"] is not None:
                st = buf["unit_stats"][k]
                o.update({"not_best_count": int(st[0]), "cnt_unassigned": int(st[1]), "n_windows": int(st[2]),
                          "repaired_windows": int(st[3]), "budget_windows": int(st[4]), "search_nodes": int(st[5]),
                          "dp_windows": int(st[6]), "dfs_components": int(st[7])})
            out.append(o)
        return out

    # ------------------------------------------------------------------------------------------
    def scale_load(self, factors, trace_rank=None):
        """helpers/transforms.py:10-40 (repeat_change_spans) on the resident table (tw_scale_load): per unit an integer load
        factor; trace_rank = per unit [n_in] ranks of the requests' trace ids (tie order of the re-sort; None = current
        order).  Needs set_truth.  Every call scales the table as uploaded.  Returns per unit (in_perm [n_in],
        [out_perm_e [n_in]] * E, time_scale) -- old index of every new position.  (self.units keeps the arrays as uploaded.)"""
        f = np.ascontiguousarray(np.asarray(factors, dtype=np.int32))
        if len(f) != len(self.units):
            raise ValueError("one load factor per unit")
        tr = None
        if trace_rank is not None:
            tr = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a in trace_rank]), dtype=np.int32)
            if len(tr) != self._in_off[-1]:
                raise ValueError("trace_rank does not match the loaded batch")
        n_out = int(sum(int(u.out_off[-1]) for u in self.units))
        ip = np.zeros(int(self._in_off[-1]), dtype=np.int32)
        op = np.zeros(n_out, dtype=np.int32)
        ts = np.zeros(len(self.units), dtype=np.float64)
        self._check(self._lib.tw_scale_load(self._h, _vp(f), _vp(tr), _vp(ip), _vp(op), _vp(ts)))
        out, o = [], 0
        for k, u in enumerate(self.units):
            a, b = int(self._in_off[k]), int(self._in_off[k + 1])
            outs = []
            for e in range(u.E):
                m = int(u.out_off[e + 1] - u.out_off[e])
                outs.append(op[o:o + m].copy())
                o += m
            out.append((ip[a:b].copy(), outs, float(ts[k])))
        return out

    def worklists(self):
        """Sizes of the device work lists of the last pass (debug aid, tw_debug_worklists): windows sent to the exact
        search, spans sent to the wavefront enumeration per endpoint count, spans whose enumeration was split into parts
        and, of those, the ones enumerated once more as a whole (order of equal scores not decided by the parts); items the lean wavefront
        kernel handed to k_enumerate_heavy, spans that were refused parts (budget of extra list entries / arena exhausted), list parts."""
        out = np.zeros(16, dtype=np.int32)
        self._lib.tw_debug_worklists.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._check(self._lib.tw_debug_worklists(self._h, _vp(out)))
        return {"select_windows": int(out[0]), "enumerate_spans": [int(x) for x in out[1:10]], "split_spans": int(out[10]), "split_redone": int(out[11]),
                "handed_to_heavy": int(out[12]), "parts_refused": int(out[13]), "list_parts": int(out[14])}

    def set_truth(self, true_parent, in_trace=None, n_traces=0):
        """Ground truth of the loaded batch: per unit [E, n_in] index of the true outgoing span; optionally per
        unit [n_in] trace numbers in [0, n_traces) for the per-trace (end-to-end) accuracy."""
        t = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a in true_parent]), dtype=np.int32)
        if len(t) != self._ie_off[-1]:
            raise ValueError("truth does not match the loaded batch")
        tr = None
        if in_trace is not None:
            tr = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a in in_trace]), dtype=np.int32)
            if len(tr) != self._in_off[-1]:
                raise ValueError("in_trace does not match the loaded batch")
        self._n_traces = int(n_traces) if tr is not None else 0
        self._check(self._lib.tw_set_truth(self._h, _vp(t), _vp(tr), int(n_traces)))

    def evaluate(self, trace_flags=False):
        """helpers/utils.py:62-145 on the device, for the last pass run.  Returns per unit
        {n_in, correct, correct_topk, unassigned, accuracy, topk_accuracy} and, when trace numbers were given,
        (traces right, traces right under top-5) -- plus the raw [2, n_traces] wrong-flags if asked for."""
        per = np.zeros((len(self.units), 4), dtype=np.int64)
        flags = np.zeros((2, self._n_traces), dtype=np.uint8) if (trace_flags and self._n_traces) else None
        e2e = np.zeros(2, dtype=np.int64) if self._n_traces else None
        self._check(self._lib.tw_evaluate(self._h, _vp(per), _vp(flags), _vp(e2e)))
        out = [{"n_in": int(r[0]), "correct": int(r[1]), "correct_topk": int(r[2]), "unassigned": int(r[3]),
                "accuracy": float(r[1]) / max(int(r[0]), 1), "topk_accuracy": float(r[2]) / max(int(r[0]), 1)} for r in per]
        if e2e is None:
            return out
        return (out, (int(e2e[0]), int(e2e[1])), flags) if trace_flags else (out, (int(e2e[0]), int(e2e[1])))

    # ------------------------------------------------------------------------------------------
    def baseline(self, kind):
        """The reference's FCFS ("FCFS", algorithms/fcfs.py) or vPath ("vPath", algorithms/vpath.py; needs set_truth) on the
        resident table (tw_run_baseline).  Per unit parent [E, n_in] int32, -1 = ("NA", "NA")."""
        code = {"FCFS": 4, "vPath": 7}[kind]
        par = np.empty(int(self._ie_off[-1]), dtype=np.int32)
        self._check(self._lib.tw_run_baseline(self._h, code, _vp(par)))
        return [par[int(self._ie_off[k]):int(self._ie_off[k + 1])].reshape(u.E, u.n_in) for k, u in enumerate(self.units)]

    def wap5(self, ep_names, state=None):
        """The reference's WAP5 (algorithms/wap5.py:257-351) on the resident table: tw_wap5_delays -> the delay samples
        added up per callee name in service order (the reference keeps one predictor object for a run; `state` = the dict
        {name: [sum in microseconds, count]} carried from batch to batch) -> tw_wap5_parents.  ep_names: per unit the
        endpoint names in the unit's endpoint order.  Returns (per unit parent [E, n_in]: -1 none, -2 several calls for one
        request -- counted as wrong like the reference does --, state, per unit options[e][i] = the calls given to request i,
        the reference's list-valued assignment)."""
        from fractions import Fraction

        state = {} if state is None else state
        nu = len(self.units)
        sums = np.zeros((nu, _ffi.TW_MAX_EP), dtype=np.int64)
        cnts = np.zeros((nu, _ffi.TW_MAX_EP), dtype=np.int32)
        self._check(self._lib.tw_wap5_delays(self._h, _vp(sums), _vp(cnts), None))
        mean = np.zeros((nu, _ffi.TW_MAX_EP), dtype=np.float64)
        for k, u in enumerate(self.units):
            scale = Fraction(1) if u.time_scale is None else Fraction(u.time_scale)
            for e in np.argsort(u.key_rank, kind="stable"):   # for out_ep in out_span_partitions.keys()
                st = state.setdefault(ep_names[k][int(e)], [Fraction(0), 0])
                st[0] += int(sums[k, e]) * scale
                st[1] += int(cnts[k, e])
                if st[1] == 0:
                    raise ValueError("WAP5: no delay sample for endpoint %r (statistics.mean of an empty list raises in the reference)" % ep_names[k][int(e)])
                mean[k, e] = float(st[0] / st[1])               # statistics.mean: exact, rounded once
        par = np.empty(int(self._ie_off[-1]), dtype=np.int32)
        req = np.empty(int(sum(int(u.out_off[-1]) for u in self.units)), dtype=np.int32)
        self._check(self._lib.tw_wap5_parents(self._h, _vp(mean), _vp(par), _vp(req)))
        options, o = [], 0
        for u in self.units:
            per = []
            for e in range(u.E):
                m = int(u.out_off[e + 1] - u.out_off[e])
                lists = [[] for _ in range(u.n_in)]
                for x in np.flatnonzero(req[o:o + m] >= 0):
                    lists[int(req[o + x])].append(int(x))
                per.append(lists)
                o += m
            options.append(per)
        return [par[int(self._ie_off[k]):int(self._ie_off[k + 1])].reshape(u.E, u.n_in) for k, u in enumerate(self.units)], state, options

    def find_order(self, units, true_parent):
        """executor.py:214-285 (FindOrder) on the device for units that need not be loaded (endpoints in any
        order): per unit the uint8 [E, E] call-order relation implied by the true assignments."""
        units = list(units)
        in_off = np.zeros(len(units) + 1, dtype=np.int64)
        np.cumsum([u.n_in for u in units], out=in_off[1:])
        unit_E = np.array([u.E for u in units], dtype=np.int32)
        ep_off, base = [0], 0
        for u in units:
            ep_off.extend((base + u.out_off[1:]).tolist())
            base += int(u.out_off[-1])
        ep_off = np.array(ep_off, dtype=np.int64)
        os_ = np.ascontiguousarray(np.concatenate([u.out_start for u in units]))
        oe_ = np.ascontiguousarray(np.concatenate([u.out_end for u in units]))
        t = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32).ravel() for a in true_parent]), dtype=np.int32)
        dag = np.zeros(int(sum(int(e) * int(e) for e in unit_E)), dtype=np.uint8)
        self._check(self._lib.tw_find_order(self._h, len(units), _vp(in_off), _vp(unit_E), _vp(ep_off), _vp(os_), _vp(oe_), _vp(t), _vp(dag)))
        out, p = [], 0
        for e in unit_E:
            out.append(dag[p:p + int(e) * int(e)].reshape(int(e), int(e)).copy())
            p += int(e) * int(e)
        return out

    def build_distributions(self, start, dur, ep, large_delay, E):
        """The sweep of BuildDistributions (traceweaver_v3.py:120-169) on the device: per span of the merged, start-ordered
        list the pair index a * (E + 1) + b its delay sample belongs to (-1 none) and the sample."""
        start = np.ascontiguousarray(start, dtype=np.int64)
        dur = np.ascontiguousarray(dur, dtype=np.int64)
        ep = np.ascontiguousarray(ep, dtype=np.uint8)
        key = np.empty(len(start), dtype=np.int32)
        val = np.empty(len(start), dtype=np.int64)
        self._check(self._lib.tw_build_distributions(self._h, len(start), _vp(start), _vp(dur), _vp(ep), int(large_delay), int(E), _vp(key), _vp(val)))
        return key, val

    def hbm_copy_gbps(self, nbytes=1 << 30, iters=10):
        """Measured HBM rate of a plain streaming copy kernel on this device (read + written GB/s)."""
        g = ctypes.c_double(0.0)
        self._check(self._lib.tw_measure_hbm_copy(self._h, int(nbytes), int(iters), ctypes.byref(g)))
        return float(g.value)

    def timing(self):
        """ms of the last pass: total, enumerate kernel, select kernel, windows, claim+detect+repair, params (HIP events); the last
        refit; repair rounds; host wall clock of submitting the first enumeration and of the whole pass call."""
        ms = np.zeros(10, dtype=np.float64)
        self._check(self._lib.tw_get_timing(self._h, _vp(ms), 10))
        return dict(zip(("pass", "enumerate", "select", "windows", "repair", "params", "fit", "rounds", "host_enum_submit", "host_pass"), ms.tolist()))
