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
