"""CPU: the C-ABI library loads and exports every symbol include/wm.h declares; the ctypes
signature table matches the header; the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wm.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char \*)\s*(wm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        nargs = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        out[m.group(2)] = nargs
    return out


def test_header_declares_the_expected_entry_points():
    fns = header_functions()
    for name in ("wm_conv2d_fwd_f32", "wm_conv2d_bwd_data_f32", "wm_conv2d_bwd_wb_f32",
                 "wm_conv2d_bwd_wb_workspace_bytes", "wm_gram_f32", "wm_gram_workspace_bytes",
                 "wm_normal_solve_f64", "wm_last_error", "wm_version"):
        assert name in fns


def test_library_exports_every_declared_symbol():
    from weather_model_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libwm.so not built: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), f"{name} declared in wm.h but not exported by libwm.so"


def test_ctypes_table_matches_header():
    from weather_model_amd import _lib
    fns = header_functions()
    assert set(fns) == set(_lib.SIGNATURES), set(fns) ^ set(_lib.SIGNATURES)
    for name, nargs in fns.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, name


def test_version_and_error_text_without_gpu():
    from weather_model_amd import _lib
    L = _lib.lib()
    assert L.wm_version() >= 100
    # argument validation happens before any device work, so it is checkable on CPU
    rc = L.wm_conv2d_fwd_f32(0, 0, 0, 0, 1, 1, 4, 4, 1, 3, 3, 0, 0)
    assert rc == -1
    assert b"null pointer" in L.wm_last_error()
    rc = L.wm_gram_f32(0, 0, 10, 16, 1, 0, 0, 0, 0)
    assert rc == -1
    # the volume-reading fit passes: the walk is validated on the host before anything is launched
    buf = ctypes.create_string_buffer(64)
    vol = ctypes.cast(buf, ctypes.c_void_p)
    strides = (ctypes.c_int64 * 5)(1, 1, 1, 1, 1)
    walk = (3, 6, 16, 16, 4, 0, 0, 0, 0, 2, 2, 2)         # d_mid, d_end, rows, batch, length, index0, r0-2, dims
    assert L.wm_gram_walk_f32(vol, 80, 5, 2, 2, 2, strides, *walk, vol, vol, 64, 0) < 0
    assert b"1..4 variables" in L.wm_last_error()
    assert L.wm_gram_walk_f32(vol, 60, 1, 2, 2, 2, strides, *walk, vol, vol, 64, 0) < 0
    assert b"resident files" in L.wm_last_error()
    assert L.wm_linear_metrics_walk_f32(vol, 80, 1, 2, 2, 2, None, *walk, vol, 1, 1e-7, vol, vol, 64, 0) < 0
    assert b"null pointer" in L.wm_last_error()
    bad = (3, 6, 16, 16, 4, 0, 0, 0, 0, 3, 2, 2)          # walk region beyond the grid
    assert L.wm_linear_metrics_walk_f32(vol, 80, 1, 2, 2, 2, strides, *bad, vol, 1, 1e-7, vol, vol, 64, 0) < 0
    assert b"leaves the grid" in L.wm_last_error()
    assert L.wm_gram_walk_workspace_bytes(7) == 0


def test_ops_refuse_cpu_tensors():
    from weather_model_amd import ops
    x = torch.zeros(1, 1, 5, 5)
    w = torch.zeros(1, 1, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d(x, w)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gram(torch.zeros(4, 2), torch.zeros(4, 1))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "weather_model_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libwm_oracle" not in text, f


def test_exchange_entry_points_refuse_use_without_a_communicator():
    """wm_comm_* (the C-ABI exchange step): argument checks and the no-communicator state are
    checkable without a GPU; librccl is only opened by the calls that need it."""
    from weather_model_amd import _lib
    L = _lib.lib()
    assert L.wm_comm_size() == 0
    assert L.wm_allreduce_sum(0, 0, 7, 0) == -1 and b"dtype" in L.wm_last_error()
    assert L.wm_allreduce_sum(0, 4, 0, 0) == -1 and b"null buffer" in L.wm_last_error()
    buf = ctypes.create_string_buffer(64)
    assert L.wm_allreduce_sum(ctypes.cast(buf, ctypes.c_void_p), 4, 0, 0) == -1
    assert b"no communicator" in L.wm_last_error()
    assert L.wm_comm_init(2, 2, ctypes.cast(buf, ctypes.c_void_p)) == -1      # rank outside world
    assert L.wm_comm_init(0, 1, 0) == -1
    assert L.wm_comm_destroy() == 0                                           # nothing to destroy
