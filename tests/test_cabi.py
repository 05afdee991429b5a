"""The C-ABI library loads and exports every function include/traceweaver_amd.h declares (no compute)."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "traceweaver_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tw_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ("tw_create", "tw_destroy", "tw_last_error", "tw_load_batch", "tw_run_pass1", "tw_get_gaps",
                 "tw_set_mixtures", "tw_run_pass2", "tw_get_results", "tw_assign_service"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from traceweaver_amd import _ffi, build

    path = build.build()
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), "libtwgpu.so does not export " + name
    assert sorted(_ffi.EXPORTS) == _declared()


def test_missing_library_fails_loudly(tmp_path):
    import pytest

    from traceweaver_amd import _ffi

    with pytest.raises(ImportError):
        _ffi.load(str(tmp_path / "nope.so"))


def test_ctypes_structures_match_the_header(tmp_path):
    """sizeof / offsetof of every structure that crosses the ABI, as a C compiler sees include/traceweaver_amd.h,
    against the ctypes declarations the Python side uses."""
    import subprocess

    from traceweaver_amd import _ffi

    fields = {"tw_batch": ("Batch", ["n_units", "unit_in_off", "key_rank", "in_start", "out_end", "batch_size", "topk", "unit_time_scale", "skip", "unit_part"]),
              "tw_results": ("Results", ["parent", "topk_score", "unit_stats"]),
              "tw_span_table": ("SpanTable", ["trace", "start", "kind"]),
              "tw_unit_set": ("UnitSet", ["n_units", "unit_in_off", "out_end", "true_child", "unit_order", "n_traces", "skipped"])}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "traceweaver_amd.h"', 'int main(void) {']
    for c_name, (_, names) in fields.items():
        src.append('printf("%s %%zu", sizeof(%s));' % (c_name, c_name))
        for f in names:
            src.append('printf(" %%zu", offsetof(%s, %s));' % (c_name, f))
        src.append('printf("\\n");')
    src += ["return 0;", "}"]
    (tmp_path / "abi.c").write_text("\n".join(src))
    exe = str(tmp_path / "abi")
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(tmp_path / "abi.c"), "-o", exe])
    for line in subprocess.check_output([exe], text=True).splitlines():
        c_name, size, *offs = line.split()
        cls = getattr(_ffi, fields[c_name][0])
        assert ctypes.sizeof(cls) == int(size), c_name
        assert [getattr(cls, f).offset for f in fields[c_name][1]] == [int(o) for o in offs], c_name


def test_loading_the_library_leaves_the_process_environment_alone(monkeypatch):
    """The engine's class streams want hardware queues of their own (csrc/tw_engine.hip, tw_create), but GPU_MAX_HW_QUEUES holds
    for every HIP user of the process: loading the library exports it only when the host opts in (TW_SET_HW_QUEUES=1), and never
    over a value the user chose.  (bench.py and the executor's command line export it themselves.)"""
    from traceweaver_amd import _ffi, build

    path = build.build()
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.delenv("TW_SET_HW_QUEUES", raising=False)
    _ffi.load(path)
    assert "GPU_MAX_HW_QUEUES" not in os.environ
    monkeypatch.setenv("TW_SET_HW_QUEUES", "1")
    _ffi.load(path)
    assert os.environ.get("GPU_MAX_HW_QUEUES") == "12"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    _ffi.load(path)
    assert os.environ.get("GPU_MAX_HW_QUEUES") == "6"
