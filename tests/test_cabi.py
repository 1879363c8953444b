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
    rc = lib.bl_embed_subtoken_max_bwd_sorted(None, 0, None, None, None, 3, None, 6, 64, hip_ops.Dropout(0.0, 0, 0).c(), 0, None, None)
    assert rc != 0 and b"null" in lib.bl_last_error()
    rc = lib.bl_segment_max_fwd(None, 0, None, None, 5, 1024, 0, None, None, None, None, 1e-5, None, None, None, None, None, None, None)
    assert rc != 0

    # the wide row GEMM rejects the shapes its tile cannot take, with the rule in the message
    rows.width[0] = 64
    rc = lib.bl_gemm_rows_x6w(ctypes.byref(rows), None, 0, ctypes.cast(buf, ctypes.c_void_p), 0, None, None, 1, 8, 128, 64,
                              ctypes.cast(buf, ctypes.c_void_p), 128, None)
    assert rc != 0 and b"multiple of 256" in lib.bl_last_error()


def test_layer_weight_image_rule_is_a_pure_function_of_the_shape(built_lib):
    """Which weight image a fused layer call expects (bl_mp_layer_weight_image): the wide row GEMM's where the layer's GEMM of
    that direction has a multiple of 256 output columns and K >= 256, the tiled one otherwise; the size query follows; the
    measurement switch turns the wide form off.  Host-side only -- no GPU."""
    from buglab.models import hip_ops

    lib = hip_ops.load_library()
    # the message GEMMs' operand split decides first: f16x3 (the default) has one image for every shape
    was = hip_ops.set_msg_gemm_mode("f16x3")
    try:
        assert [lib.bl_mp_layer_weight_image(a, b, d) for a, b in ((128, 128), (256, 256), (96, 160)) for d in (0, 1)] == [2] * 6
        assert lib.bl_mp_layer_packed_weight_elems(16, 128, 128, 0) == lib.bl_packed_weight_elems_h3(16, 256, 128) == 16 * (256 // 32) * 8192
        assert lib.bl_mp_layer_packed_weight_elems(16, 256, 256, 1) == lib.bl_packed_weight_elems_h3(16, 256, 512)
        assert hip_ops.msg_gemm_mode() == "f16x3" and hip_ops.set_msg_gemm_mode("bf16x6") == "f16x3"
        _check_bf16x6_image_rule(lib, hip_ops)
    finally:
        hip_ops.set_msg_gemm_mode(was)


def _check_bf16x6_image_rule(lib, hip_ops):
    assert lib.bl_set_rows_tile(256) in (128, 256)
    # (Din, Dm): forward GEMM N = Dm, K = 2 Din; input-gradient GEMM N = 2 Din, K = Dm
    assert [lib.bl_mp_layer_weight_image(128, 128, d) for d in (0, 1)] == [0, 0]   # hidden-128 plain layer: Dm = 128 / K = 128
    assert [lib.bl_mp_layer_weight_image(256, 256, d) for d in (0, 1)] == [1, 1]   # concat layer at hidden 128, plain layer at 256
    assert [lib.bl_mp_layer_weight_image(512, 512, d) for d in (0, 1)] == [1, 1]
    assert [lib.bl_mp_layer_weight_image(96, 160, d) for d in (0, 1)] == [0, 0]
    T = 16
    assert lib.bl_mp_layer_packed_weight_elems(T, 256, 256, 0) == lib.bl_packed_weight_elems_x6w(T, 512, 256)
    assert lib.bl_mp_layer_packed_weight_elems(T, 128, 128, 0) == T * 1 * (256 // 32) * 12288
    assert lib.bl_packed_weight_elems_x6w(1, 512, 256) == (512 // 32) * 24576   # one 48 KB block per 256 columns and 32-k stage
    prev = lib.bl_set_rows_tile(128)
    try:
        assert prev == 256 and lib.bl_mp_layer_weight_image(256, 256, 0) == 0 and not hip_ops.rows_x6w_ok(256, 512)
    finally:
        lib.bl_set_rows_tile(prev)
    assert hip_ops.rows_x6w_ok(256, 512) and not hip_ops.rows_x6w_ok(256, 128) and not hip_ops.rows_x6w_ok(128, 512)


def test_ctypes_structures_match_the_c_layout(tmp_path):
    """Every structure that crosses the boundary by value or by pointer: size and every field offset as gcc lays the header's
    declaration out == what the ctypes mirror in hip_ops/_lib.py says (a mismatch would shift every later field silently)."""
    import ctypes
    import importlib

    L = importlib.import_module("buglab.models.hip_ops._lib")  # (hip_ops._lib itself resolves to the loaded library)

    names = ["bl_rows_t", "bl_rows_packed_t", "bl_dropout_t", "bl_mp_layer_t", "bl_pack_job_t", "bl_bug_loss_t", "bl_x6_epi_t", "bl_head_view_t",
             "bl_packed_head_view_t", "bl_great_layer_t", "bl_great_layer_grads_t"]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for n in names:
        cls = getattr(L, n)
        lines.append(f'  printf("{n} %zu", sizeof({n}));')
        for f, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({n}, {f}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == len(names)
    for line in out:
        n, size, *offs = line.split()
        cls = getattr(L, n)
        assert ctypes.sizeof(cls) == int(size), (n, ctypes.sizeof(cls), size)
        assert [getattr(cls, f).offset for f, _ in cls._fields_] == [int(o) for o in offs], n
