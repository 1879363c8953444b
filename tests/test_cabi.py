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
