"""The C-ABI library: builds for gfx950, loads, exports every symbol include/dmet.h declares, and validates
arguments without touching a GPU (no compute calls here)."""
import ctypes
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "dmet.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmet_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from deepmetv2_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("libdmet_hip.so not built and no hipcc here")
        build.build_hip()
    return _lib.load()


def test_header_and_binding_agree():
    from deepmetv2_amd import _lib
    assert _header_functions() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_header_symbol(lib):
    for name in _header_functions():
        assert hasattr(lib, name), f"{name} declared in include/dmet.h but not exported"


def test_version_and_argument_validation(lib):
    assert lib.dmet_version() == 1
    assert lib.dmet_device_available() in (0, 1)
    # bad arguments are rejected before any HIP call: -EINVAL and a message
    rc = lib.dmet_knn_f32(None, None, 1, 10, 32, 0, None, None, None, 0, None)
    assert rc == -22 and b"k=0" in lib.dmet_last_error()
    rc = lib.dmet_knn_f32(None, None, 1, 10, 200, 4, None, None, None, 0, None)
    assert rc == -22 and b"D=200" in lib.dmet_last_error()
    rc = lib.dmet_gather_max_f32(None, None, None, None, 0, 10, 300, 32, None, None, None)
    assert rc == -22
    rc = lib.dmet_xty_f32(None, None, 10, 100, 4, None, None, 0, None)
    assert rc == -22
    assert lib.dmet_knn_workspace_bytes(1000, 2, 32, 16) > 1000 * 16 * 8
    assert lib.dmet_edgeconv_linear_workspace_bytes(1000, 32) >= 2 * 1000 * 32 * 4
    # empty problems are no-ops
    assert lib.dmet_knn_f32(None, None, 0, 0, 32, 4, None, None, None, 0, None) == 0


def test_oracle_is_not_reachable_from_the_product():
    """The shipped package must not import or link the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "deepmetv2_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "libdmet_oracle" not in text and "ref_ops" not in text, f"{f} reaches into oracle/"
                assert not re.search(r"#include\s+[\"<][^\"<>]*oracle", text), f"{f} includes oracle sources"


def test_deferral_and_size_hint_host_state(lib):
    """The host-side state machines of dmet_finalize_defer_begin / _pending / _flush and dmet_knn_size_hint need no GPU:
    outside a deferral nothing is pending (-1), inside it an empty queue flushes without a launch (stream NULL), a second
    flush is a no-op, negative sizes are refused through the error string like every other argument check."""
    assert lib.dmet_finalize_pending() == -1
    assert lib.dmet_finalize_flush(None) == 0            # no deferral: nothing to do
    assert lib.dmet_finalize_defer_begin() == 0
    assert lib.dmet_finalize_pending() == 0
    assert lib.dmet_finalize_defer_begin() == 0          # beginning again resets the queue
    assert lib.dmet_finalize_flush(None) == 0            # empty queue: no launch
    assert lib.dmet_finalize_pending() == -1
    assert lib.dmet_knn_size_hint(800, 4500) == 0
    assert lib.dmet_knn_size_hint(0, 0) == 0             # "unknown" clears it
    assert lib.dmet_knn_size_hint(-1, 10) != 0
    assert b"dmet_knn_size_hint" in lib.dmet_last_error()
