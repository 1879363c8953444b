"""CPU: the C-ABI shared library loads without a GPU and exports exactly what include/buglab_hip.h
declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

from tests.conftest import PKG, ROOT

HEADER = os.path.join(ROOT, "include", "buglab_hip.h")
LIB = os.path.join(PKG, "buglab", "models", "hip_ops", "libbuglab_hip.so")


def _header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bl_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g

        g.build()
    return LIB


def test_header_matches_exports_and_ctypes_table(built_lib):
    from buglab.models import hip_ops

    declared = _header_symbols()
    nm = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (bl_[a-z0-9_]+)", nm)))
    assert declared == exported, (set(declared) ^ set(exported))
    assert sorted(hip_ops.EXPORTED_SYMBOLS) == declared
    lib = hip_ops.load_library()  # dlopen works without a GPU; prototypes resolve
    assert lib.bl_version() >= 1 and lib.bl_last_error() is not None


def test_header_cites_the_reference_for_every_entry_point():
    src = open(HEADER).read()
    assert "extern \"C\"" in src
    for cite in ("gnnlayerdefs.py", "localizationmodule.py", "fixermodules.py", "utils.py:15-28", "train.py:104", "modelregistry.py:59-82"):
        assert cite in src, cite


def test_argument_errors_are_reported_without_touching_the_gpu(built_lib):
    """Every entry point validates its arguments before the first HIP call: a bad call returns BL_EINVAL
    (non-zero) and leaves a message for bl_last_error() -- the library never falls back or guesses."""
    import ctypes

    from buglab.models import hip_ops

    lib = hip_ops.load_library()
    rows = hip_ops.bl_rows_packed_t()
    rows.nsrc = 1
    rows.width[0] = 48  # not a multiple of 32
    buf = (ctypes.c_uint16 * 64)()
    rows.xp[0] = ctypes.cast(buf, ctypes.c_void_p).value
    rc = lib.bl_gemm_rows_x6(ctypes.byref(rows), None, 0, ctypes.cast(buf, ctypes.c_void_p), 0, None, None, 1, 8, 32, 48,
                             ctypes.cast(buf, ctypes.c_void_p), 32, None)
    assert rc != 0 and b"multiple of 32" in lib.bl_last_error()
    rc = lib.bl_pack_weights_x6(ctypes.cast(buf, ctypes.c_void_p), 1, 40, 16, 1, ctypes.cast(buf, ctypes.c_void_p), None)
    assert rc != 0 and b"multiple of 32" in lib.bl_last_error()
    rc = lib.bl_layernorm_bwd(None, None, None, None, None, 4, 8, None, None, None, None, None, None)
    assert rc != 0 and b"null" in lib.bl_last_error()
    rc = lib.bl_embed_subtoken_max_bwd_sorted(None, 0, None, None, None, 3, None, 6, 64, hip_ops.Dropout(0.0, 0, 0).c(), None, None)
    assert rc != 0 and b"null" in lib.bl_last_error()
    rc = lib.bl_segment_max_fwd(None, 0, None, None, 5, 1024, 0, None, None, None, None, 1e-5, None, None, None, None, None, None, None)
    assert rc != 0
