"""ctypes declarations of include/traceweaver_amd.h.  The shared library must have been built by
`__graft_entry__.build()` (hipcc, gfx950); there is no fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libtwgpu.so")

TW_MAX_EP = 8
TW_TOPK = 5
TW_MAX_COMP = 5

STATUS = {0: "TW_OK", -1: "TW_ERR_ARG", -2: "TW_ERR_UNSUPPORTED", -3: "TW_ERR_DEVICE", -4: "TW_ERR_STATE",
          -5: "TW_ERR_WINDOW_WIDTH", -6: "TW_ERR_WINDOW_SIZE", -7: "TW_ERR_NAN_PARAMS", -8: "TW_ERR_SKIP_PARAMS",
          -9: "TW_ERR_SKIP_REFERENCE_RAISES"}
TW_SKIP_BASE = 1024
TW_SKIP_STRIDE = 128

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)


class Batch(ctypes.Structure):
    _fields_ = [
        ("n_units", ctypes.c_int32),
        ("unit_in_off", ctypes.c_void_p), ("unit_E", ctypes.c_void_p), ("ep_off", ctypes.c_void_p),
        ("dag", ctypes.c_void_p), ("key_rank", ctypes.c_void_p),
        ("in_start", ctypes.c_void_p), ("in_end", ctypes.c_void_p),
        ("out_start", ctypes.c_void_p), ("out_end", ctypes.c_void_p),
        ("batch_size", ctypes.c_int32), ("batch_size_mis", ctypes.c_int32), ("topk", ctypes.c_int32),
        ("unit_time_scale", ctypes.c_void_p),
        ("skip", ctypes.c_void_p),
        ("unit_part", ctypes.c_void_p),
    ]


class DeviceView(ctypes.Structure):
    _fields_ = [("parent", ctypes.c_void_p), ("parent_count", ctypes.c_int64), ("gaps", ctypes.c_void_p), ("gaps_count", ctypes.c_int64),
                ("device", ctypes.c_int32)]


class SkipUnit(ctypes.Structure):
    _fields_ = [("n_tw", ctypes.c_int32), ("tw_start", ctypes.c_void_p), ("pool", ctypes.c_void_p), ("dist", ctypes.c_void_p)]


class SpanTable(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("trace", "span_id", "service", "op_name", "parent", "start", "duration", "kind")]


class UnitSet(ctypes.Structure):
    _fields_ = [("n_units", ctypes.c_int32)] + [(k, ctypes.c_void_p) for k in (
        "unit_in_off", "unit_E", "ep_off", "dag", "key_rank", "in_start", "in_end", "out_start", "out_end",
        "true_child", "in_trace", "in_row", "out_row", "unit_service", "ep_name", "in_ep_name", "unit_order")] + [
        ("n_traces", ctypes.c_int64), ("skipped", ctypes.c_int32 * 4)]


class Results(ctypes.Structure):
    _fields_ = [
        ("parent", ctypes.c_void_p), ("topk_idx", ctypes.c_void_p), ("topk_score", ctypes.c_void_p),
        ("topk_n", ctypes.c_void_p), ("chosen", ctypes.c_void_p), ("leaves", ctypes.c_void_p),
        ("window_end", ctypes.c_void_p), ("unit_stats", ctypes.c_void_p),
    ]


EXPORTS = ["tw_create", "tw_destroy", "tw_last_error", "tw_load_batch", "tw_run_pass1", "tw_get_gaps", "tw_set_gaps",
           "tw_device_buffers", "tw_set_gaps_device", "tw_set_mixtures", "tw_fit_mixtures", "tw_fit_mixtures_seeded", "tw_set_fit_seed", "tw_fit_rows", "tw_fit_mixtures_tape", "tw_get_mixtures", "tw_run_pass2", "tw_get_results", "tw_get_gauss_params", "tw_get_timing",
           "tw_assign_service", "tw_find_order", "tw_set_truth", "tw_evaluate", "tw_measure_hbm_copy", "tw_host_alloc", "tw_host_free", "tw_build_distributions", "tw_scale_load", "tw_run_baseline", "tw_wap5_delays", "tw_wap5_parents",
           "tw_corpus_create", "tw_corpus_destroy", "tw_corpus_last_error", "tw_corpus_add_files", "tw_corpus_set_callers", "tw_corpus_counts",
           "tw_corpus_string", "tw_corpus_loop_origin", "tw_corpus_trace_names", "tw_corpus_span_table", "tw_corpus_build_units"]


def load(path=None):
    path = path or DEFAULT_LIB
    # Hardware queues for the engine's class streams (tw_create has the measurements).  GPU_MAX_HW_QUEUES is read by the HIP
    # runtime when it initialises and holds for every HIP user of the process (torch, RCCL): loading this library does not
    # change it unless the host opts in with TW_SET_HW_QUEUES=1 (the applications of this repository -- bench.py, the
    # executor's command line -- export it themselves before anything touches the device).
    if os.environ.get("TW_SET_HW_QUEUES", "0") not in ("", "0"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    if not os.path.exists(path):
        raise ImportError(
            "traceweaver_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    lib.tw_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.tw_destroy.argtypes = [vp]
    lib.tw_destroy.restype = None
    lib.tw_last_error.argtypes = [vp]
    lib.tw_last_error.restype = ctypes.c_char_p
    lib.tw_load_batch.argtypes = [vp, ctypes.POINTER(Batch), ctypes.c_int]
    lib.tw_run_pass1.argtypes = [vp]
    lib.tw_get_gaps.argtypes = [vp, vp]
    lib.tw_set_gaps.argtypes = [vp, vp]
    lib.tw_run_baseline.argtypes = [vp, ctypes.c_int, vp]
    lib.tw_wap5_delays.argtypes = [vp, vp, vp, vp]
    lib.tw_wap5_parents.argtypes = [vp, vp, vp, vp]
    lib.tw_device_buffers.argtypes = [vp, ctypes.POINTER(DeviceView)]
    lib.tw_set_gaps_device.argtypes = [vp, vp]
    lib.tw_scale_load.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.tw_set_mixtures.argtypes = [vp, vp, vp]
    lib.tw_run_pass2.argtypes = [vp]
    lib.tw_fit_mixtures.argtypes = [vp]
    lib.tw_fit_mixtures_seeded.argtypes = [vp, vp]
    lib.tw_set_fit_seed.argtypes = [vp, ctypes.c_uint32]
    lib.tw_fit_rows.argtypes = [vp, vp]
    lib.tw_fit_mixtures_tape.argtypes = [vp, vp, ctypes.c_int64, vp]
    lib.tw_get_mixtures.argtypes = [vp, vp, vp]
    lib.tw_get_results.argtypes = [vp, ctypes.c_int, ctypes.POINTER(Results)]
    lib.tw_get_gauss_params.argtypes = [vp, vp]
    lib.tw_get_timing.argtypes = [vp, vp, ctypes.c_int32]
    lib.tw_assign_service.argtypes = [vp, ctypes.c_int32, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp,
                                      ctypes.POINTER(Results)]
    lib.tw_find_order.argtypes = [vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.tw_set_truth.argtypes = [vp, vp, vp, ctypes.c_int64]
    lib.tw_evaluate.argtypes = [vp, vp, vp, vp]
    lib.tw_measure_hbm_copy.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    lib.tw_build_distributions.argtypes = [vp, ctypes.c_int64, vp, vp, vp, ctypes.c_int64, ctypes.c_int32, vp, vp]
    lib.tw_host_alloc.argtypes = [ctypes.c_int64, ctypes.POINTER(vp)]
    lib.tw_host_free.argtypes = [vp]
    lib.tw_host_free.restype = None
    lib.tw_corpus_create.argtypes = [ctypes.POINTER(vp)]
    lib.tw_corpus_destroy.argtypes = [vp]
    lib.tw_corpus_destroy.restype = None
    lib.tw_corpus_last_error.argtypes = [vp]
    lib.tw_corpus_last_error.restype = ctypes.c_char_p
    lib.tw_corpus_add_files.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    lib.tw_corpus_set_callers.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32]
    lib.tw_corpus_counts.argtypes = [vp, vp]
    lib.tw_corpus_string.argtypes = [vp, ctypes.c_int32]
    lib.tw_corpus_string.restype = ctypes.c_char_p
    lib.tw_corpus_loop_origin.argtypes = [vp, ctypes.c_char_p]
    lib.tw_corpus_loop_origin.restype = ctypes.c_char_p
    lib.tw_corpus_trace_names.argtypes = [vp, vp]
    lib.tw_corpus_span_table.argtypes = [vp, ctypes.POINTER(SpanTable)]
    lib.tw_corpus_build_units.argtypes = [vp, ctypes.POINTER(UnitSet)]
    for name in EXPORTS:
        if name not in ("tw_destroy", "tw_last_error", "tw_host_free", "tw_corpus_destroy", "tw_corpus_last_error", "tw_corpus_string", "tw_corpus_loop_origin"):
            getattr(lib, name).restype = ctypes.c_int
    return lib
