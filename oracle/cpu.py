"""ctypes front-end of the C oracle (``wm_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Every function takes float32 numpy arrays (what the product sees), computes in float64 and
returns float64 arrays.  See ``wm_oracle.c`` for the reference file:line each one follows and
for the pinning status (conv: pinned by tests/golden; regression: parity unpinned).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwm_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile wm_oracle.c with gcc (no GPU needed)."""
    src = os.path.join(_HERE, "wm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libwm_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        c_int, c_i64, c_dbl = ctypes.c_int, ctypes.c_int64, ctypes.c_double
        vp = ctypes.c_void_p
        L.wm_oracle_threads.restype = c_int
        L.wm_oracle_conv2d_fwd.argtypes = [vp, vp, vp, vp] + [c_int] * 7
        L.wm_oracle_conv2d_bwd_data.argtypes = [vp, vp, vp] + [c_int] * 7
        L.wm_oracle_conv2d_bwd_wb.argtypes = [vp, vp, vp, vp] + [c_int] * 7
        L.wm_oracle_conv2d_fwd_bwd_wb_f32.argtypes = [vp] * 7 + [c_int] * 7
        L.wm_oracle_conv3d_fwd.argtypes = [vp, vp, vp, vp] + [c_int] * 9
        L.wm_oracle_conv3d_bwd_data.argtypes = [vp, vp, vp] + [c_int] * 9
        L.wm_oracle_conv3d_bwd_wb.argtypes = [vp, vp, vp, vp] + [c_int] * 9
        L.wm_oracle_maxpool3d_fwd.argtypes = [vp, vp, vp] + [c_int] * 5
        L.wm_oracle_maxpool3d_bwd.argtypes = [vp, vp, vp] + [c_int] * 5
        L.wm_oracle_gram.argtypes = [vp, vp, c_i64, c_int, c_int, vp]
        L.wm_oracle_gram_f32.argtypes = [vp, vp, c_i64, c_int, c_int, vp]
        L.wm_oracle_normal_solve.argtypes = [vp, c_int, c_int, c_int, c_dbl, vp]
        L.wm_oracle_normal_solve.restype = c_int
        _lib = L
    return _lib


def threads() -> int:
    return int(lib().wm_oracle_threads())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _conv_dims(x, w):
    N, Cin, H, W = x.shape
    Cout, Cin2, kH, kW = w.shape
    if Cin != Cin2:
        raise ValueError(f"channel mismatch: x has {Cin}, weight expects {Cin2}")
    return N, Cin, H, W, Cout, kH, kW


def conv2d_fwd(x, w, b=None):
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    N, Cin, H, W, Cout, kH, kW = _conv_dims(x, w)
    y = np.zeros((N, Cout, max(H - kH + 1, 0), max(W - kW + 1, 0)), dtype=np.float64)
    lib().wm_oracle_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), N, Cin, H, W, Cout, kH, kW)
    return y


def conv2d_bwd_data(gy, w, x_shape):
    gy, w = _f32(gy), _f32(w)
    N, Cin, H, W = x_shape
    Cout, _, kH, kW = w.shape
    gx = np.zeros((N, Cin, H, W), dtype=np.float64)
    lib().wm_oracle_conv2d_bwd_data(_p(gy), _p(w), _p(gx), N, Cin, H, W, Cout, kH, kW)
    return gx


def conv2d_bwd_wb(x, gy, w_shape):
    x, gy = _f32(x), _f32(gy)
    N, Cin, H, W = x.shape
    Cout, _, kH, kW = w_shape
    gw = np.zeros(tuple(w_shape), dtype=np.float64)
    gb = np.zeros((Cout,), dtype=np.float64)
    lib().wm_oracle_conv2d_bwd_wb(_p(x), _p(gy), _p(gw), _p(gb), N, Cin, H, W, Cout, kH, kW)
    return gw, gb


def conv2d_fwd_bwd_wb_f32(x, w, b, gy):
    """float32 CPU leg used by bench.py's cpu_baseline only."""
    x, w, b, gy = _f32(x), _f32(w), _f32(b), _f32(gy)
    N, Cin, H, W, Cout, kH, kW = _conv_dims(x, w)
    y = np.empty((N, Cout, H - kH + 1, W - kW + 1), dtype=np.float32)
    gw = np.zeros(w.shape, dtype=np.float64)
    gb = np.zeros((Cout,), dtype=np.float64)
    lib().wm_oracle_conv2d_fwd_bwd_wb_f32(_p(x), _p(w), _p(b), _p(gy), _p(y), _p(gw), _p(gb),
                                          N, Cin, H, W, Cout, kH, kW)
    return y, gw, gb


def _conv3_dims(x, w):
    B, Cin, D, H, W = x.shape
    Cout, Cin2, kD, kH, kW = w.shape
    if Cin != Cin2:
        raise ValueError(f"channel mismatch: x has {Cin}, weight expects {Cin2}")
    return B, Cin, D, H, W, Cout, kD, kH, kW


def conv3d_fwd(x, w, b=None):
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    dims = _conv3_dims(x, w)
    B, Cin, D, H, W, Cout, kD, kH, kW = dims
    y = np.zeros((B, Cout, D - kD + 1, H - kH + 1, W - kW + 1), dtype=np.float64)
    lib().wm_oracle_conv3d_fwd(_p(x), _p(w), _p(b), _p(y), *dims)
    return y


def conv3d_bwd_data(gy, w, x_shape):
    gy, w = _f32(gy), _f32(w)
    B, Cin, D, H, W = x_shape
    Cout, _, kD, kH, kW = w.shape
    gx = np.zeros((B, Cin, D, H, W), dtype=np.float64)
    lib().wm_oracle_conv3d_bwd_data(_p(gy), _p(w), _p(gx), B, Cin, D, H, W, Cout, kD, kH, kW)
    return gx


def conv3d_bwd_wb(x, gy, w_shape):
    x, gy = _f32(x), _f32(gy)
    B, Cin, D, H, W = x.shape
    Cout, _, kD, kH, kW = w_shape
    gw = np.zeros(tuple(w_shape), dtype=np.float64)
    gb = np.zeros((Cout,), dtype=np.float64)
    lib().wm_oracle_conv3d_bwd_wb(_p(x), _p(gy), _p(gw), _p(gb), B, Cin, D, H, W, Cout, kD, kH, kW)
    return gw, gb


def maxpool3d_fwd(x):
    """x (B,A1,A2,A3,C) -> (y, which) pooled over A1..A3 (window 2, stride 2)."""
    x = _f32(x)
    B, A1, A2, A3, C = x.shape
    y = np.zeros((B, A1 // 2, A2 // 2, A3 // 2, C), dtype=np.float32)
    which = np.zeros(y.shape, dtype=np.uint8)
    lib().wm_oracle_maxpool3d_fwd(_p(x), _p(y), _p(which), B, A1, A2, A3, C)
    return y, which


def maxpool3d_bwd(gy, which, x_shape):
    gy = _f32(gy)
    which = np.ascontiguousarray(which, dtype=np.uint8)
    B, A1, A2, A3, C = x_shape
    gx = np.zeros(tuple(x_shape), dtype=np.float32)
    lib().wm_oracle_maxpool3d_bwd(_p(gy), _p(which), _p(gx), B, A1, A2, A3, C)
    return gx


def gram(X, Y, acc=None, f32=False):
    """Gram of Z=[X|1|Y] in float64 (D x D, D=F+1+P); ``acc`` is a running sum."""
    X, Y = _f32(X), _f32(Y)
    if Y.ndim == 1:
        Y = Y[:, None]
    n, F = X.shape
    P = Y.shape[1]
    D = F + 1 + P
    if acc is None:
        acc = np.zeros((D, D), dtype=np.float64)
    fn = lib().wm_oracle_gram_f32 if f32 else lib().wm_oracle_gram
    fn(_p(X), _p(Y), n, F, P, _p(acc))
    return acc


def normal_solve(acc, F, P, fit_intercept=True, ridge=0.0):
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    beta = np.zeros((F + 1, P), dtype=np.float64)
    rc = lib().wm_oracle_normal_solve(_p(acc), F, P, int(bool(fit_intercept)), float(ridge), _p(beta))
    if rc != 0:
        raise np.linalg.LinAlgError("normal equations not positive definite")
    return beta


def lstsq(X, Y, fit_intercept=True):
    """The regression arbiter: numpy.linalg.lstsq in float64 on the up-cast float32 data
    (SURVEY.md 8d).  Returns (F+1, P), intercept last."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    if fit_intercept:
        A = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1)
        return np.linalg.lstsq(A, Y, rcond=None)[0]
    b = np.linalg.lstsq(X, Y, rcond=None)[0]
    return np.concatenate([b, np.zeros((1, Y.shape[1]))], axis=0)


# ---------------------------------------------------------------------------------------------
# fields: numpy restatements of the reference's host arithmetic.

def field_stats(x):
    """[min, max, sum, sumsq, count] over the non-NaN elements, float64 (data.py:59-60,76-77 take
    float(ds[var].max()/.min()/.mean()/.std()); xarray skips NaNs in float reductions)."""
    v = np.asarray(x, dtype=np.float32).reshape(-1)
    v = v[~np.isnan(v)].astype(np.float64)
    return np.array([v.min(), v.max(), v.sum(), (v * v).sum(), float(v.size)])


def affine_scale(x, offset, divisor):
    """(data - lo) / (hi - lo) exactly as numpy evaluates it for a float32 array and Python-float
    scalars (newConvolution.py:48): float32 subtraction and division."""
    x = np.asarray(x, dtype=np.float32)
    return (x - np.float32(offset)) / np.float32(divisor)


def gather_triplets(vol, shape, indices, time_offset=0, d_mid=3, d_end=6):
    """CustomDataset.__getitem__ (newConvolution.py:93-106) over a (T,Z,Y,X,V) volume."""
    St, Sz, Sy, Sx = shape
    idx = np.asarray(indices, dtype=np.int64) % (St * Sz * Sy * Sx)
    column = idx % Sx
    row = (idx // Sx) % Sy
    height = (idx // (Sx * Sy)) % Sz
    time = (idx // (Sx * Sy * Sz)) % St + time_offset
    g = lambda t: vol[t, height, row, column]                       # noqa: E731
    return np.stack((g(time), g(time + d_end)), axis=1), g(time + d_mid)
