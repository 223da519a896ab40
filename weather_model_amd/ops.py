"""Operator layer: PyTorch-ROCm tensors -> ctypes -> libwm.so (hand-written gfx950 HIP).

Mirrors the two library operators the reference's hot path calls:
  * ``conv2d`` / ``Conv2dFunction``  <-> torch.nn.functional.conv2d as reached from nn.Conv2d
    (newConvolution.py:129-133,149-158) and its autograd (loss.backward(), :184)
  * ``gram`` / ``normal_solve`` / ``lstsq_fit`` <-> the least-squares problem the linear Dense
    stacks minimise (multipleRegression.py:273-298, singleRegression.py:74-96)

PyTorch is plumbing only (device memory, streams).  Every op requires CUDA(HIP) tensors and
raises otherwise -- there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (weather_model_amd has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _conv_dims(x, w):
    if x.dim() != 4 or w.dim() != 4:
        raise ValueError(f"conv2d expects NCHW input and (Cout,Cin,kH,kW) weight, got "
                         f"{tuple(x.shape)} and {tuple(w.shape)}")
    N, Cin, H, W = x.shape
    Cout, Cin2, kH, kW = w.shape
    if Cin != Cin2:
        # same error class torch raises for a channel mismatch
        raise RuntimeError(f"conv2d: expected input with {Cin2} channels, got {Cin}")
    if kH > H or kW > W:
        raise RuntimeError(f"conv2d: kernel size ({kH},{kW}) can't be greater than input ({H},{W})")
    return N, Cin, H, W, Cout, kH, kW


def conv2d_forward(x, w, b=None, relu=False, out=None):
    x, w = _req(x, "input"), _req(w, "weight")
    b = None if b is None else _req(b, "bias")
    N, Cin, H, W, Cout, kH, kW = _conv_dims(x, w)
    y = _out_like(out, (N, Cout, H - kH + 1, W - kW + 1), x.device, "out")
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().wm_conv2d_fwd_f32(
            x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(), y.data_ptr(),
            N, Cin, H, W, Cout, kH, kW, int(bool(relu)), _stream()))
    return y


def _out_like(out, shape, device, name):
    if out is None:
        return torch.empty(shape, device=device, dtype=torch.float32)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_cuda or \
            not out.is_contiguous() or out.device != device:
        raise ValueError(f"{name} must be a contiguous float32 {tuple(shape)} tensor on {device}")
    return out


def conv2d_backward_data(gy, w, x_shape, relu_mask=None, out=None):
    """grad_input of the conv; with ``relu_mask`` (the post-ReLU output of the layer below, shaped
    like the result) the ReLU backward is fused in: gx = relu_mask > 0 ? gx : 0."""
    gy, w = _req(gy, "grad_output"), _req(w, "weight")
    N, Cin, H, W = (int(v) for v in x_shape)
    Cout, Cin2, kH, kW = w.shape
    if Cin2 != Cin or tuple(gy.shape) != (N, Cout, H - kH + 1, W - kW + 1):
        raise ValueError(f"grad_output {tuple(gy.shape)} / weight {tuple(w.shape)} do not belong "
                         f"to an input of shape {tuple(x_shape)}")
    gx = _out_like(out, (N, Cin, H, W), gy.device, "out")
    with torch.cuda.device(gy.device):
        if relu_mask is None:
            _lib.check(_lib.lib().wm_conv2d_bwd_data_f32(
                gy.data_ptr(), w.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout, kH, kW, _stream()))
        else:
            m = _req(relu_mask, "relu_mask")
            if tuple(m.shape) != (N, Cin, H, W):
                raise ValueError(f"relu_mask must be {(N, Cin, H, W)}, got {tuple(m.shape)}")
            _lib.check(_lib.lib().wm_conv2d_bwd_data_relu_f32(
                gy.data_ptr(), w.data_ptr(), m.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout, kH,
                kW, _stream()))
    return gx


def relu_mask_(g, y):
    """In place g = y > 0 ? g : 0 (the backward of the ReLU whose output is y)."""
    g, y = _req(g, "grad"), _req(y, "output")
    if g.shape != y.shape or not g.is_contiguous():
        raise ValueError("grad and output must be contiguous tensors of one shape")
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().wm_relu_mask_f32(g.data_ptr(), y.data_ptr(), g.numel(), _stream()))
    return g


def relu_mask(g, y, out=None):
    """out = y > 0 ? g : 0 into a separate (or new) buffer."""
    g, y = _req(g, "grad"), _req(y, "output")
    if g.shape != y.shape:
        raise ValueError("grad and output must have one shape")
    o = _out_like(out, tuple(g.shape), g.device, "out")
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().wm_relu_mask_to_f32(g.data_ptr(), y.data_ptr(), o.data_ptr(),
                                                  g.numel(), _stream()))
    return o


def conv2d_bwd_wb_workspace(x_shape, w_shape, device="cuda"):
    """A workspace tensor for ``conv2d_backward_wb(..., ws=)`` (reusable across calls of one shape)."""
    N, Cin, H, W = (int(v) for v in x_shape)
    Cout, _, kH, kW = (int(v) for v in w_shape)
    with torch.cuda.device(device):
        nbytes = _lib.lib().wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW)
    return torch.empty((max(nbytes, 8),), device=device, dtype=torch.uint8)


def conv2d_backward_wb(x, gy, w_shape, need_bias=True, ws=None, out_w=None, out_b=None):
    x, gy = _req(x, "input"), _req(gy, "grad_output")
    N, Cin, H, W = x.shape
    Cout, _, kH, kW = w_shape
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = L.wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW)
        if ws is None:
            ws = torch.empty((max(nbytes, 8),), device=x.device, dtype=torch.uint8)
        elif ws.numel() < nbytes or not ws.is_cuda or ws.dtype != torch.uint8:
            raise ValueError(f"ws must be a CUDA uint8 tensor of at least {nbytes} bytes")
        gw = _out_like(out_w, tuple(w_shape), x.device, "out_w")
        gb = _out_like(out_b, (Cout,), x.device, "out_b") if need_bias else None
        _lib.check(L.wm_conv2d_bwd_wb_f32(
            x.data_ptr(), gy.data_ptr(), gw.data_ptr(), 0 if gb is None else gb.data_ptr(),
            ws.data_ptr(), ws.numel(), N, Cin, H, W, Cout, kH, kW, _stream()))
    return gw, gb


class GradPartials:
    """The float64 partial sums a deferred weight/bias gradient leaves in its workspace
    (wm_conv2d_bwd_wb_partials_f32): consumed by ``finish()`` or ``adam_step_partials``."""

    def __init__(self, ws, nparts, entry_major, w_shape):
        self.ws, self.nparts, self.entry_major = ws, int(nparts), int(entry_major)
        self.w_shape = tuple(int(v) for v in w_shape)
        self.nw = math.prod(self.w_shape)
        self.nb = self.w_shape[0]

    def finish(self, need_bias=True, out_w=None, out_b=None):
        """The fixed-order sum: bit for bit the (gw, gb) of ``conv2d_backward_wb``."""
        dev = self.ws.device
        with torch.cuda.device(dev):
            gw = _out_like(out_w, self.w_shape, dev, "out_w")
            gb = _out_like(out_b, (self.nb,), dev, "out_b") if need_bias else None
            _lib.check(_lib.lib().wm_partials_finish_f32(
                self.ws.data_ptr(), self.nparts, self.entry_major, self.nw, self.nb, gw.data_ptr(),
                0 if gb is None else gb.data_ptr(), _stream()))
        return gw, gb


def conv2d_backward_wb_partials(x, gy, w_shape, ws=None):
    """Deferred-finalize form of ``conv2d_backward_wb`` (newConvolution.py:184-186): the partial-sum
    pass only; the consumer that runs next (``adam_step_partials`` / ``GradPartials.finish``) adds
    the partials in the fixed order.  Shapes on the MFMA kernels raise WmError (unsupported)."""
    x, gy = _req(x, "input"), _req(gy, "grad_output")
    N, Cin, H, W = x.shape
    Cout, _, kH, kW = w_shape
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = L.wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW)
        if ws is None:
            ws = torch.empty((max(nbytes, 8),), device=x.device, dtype=torch.uint8)
        elif ws.numel() < nbytes or not ws.is_cuda or ws.dtype != torch.uint8:
            raise ValueError(f"ws must be a CUDA uint8 tensor of at least {nbytes} bytes")
        layout = (ctypes.c_int * 2)()
        _lib.check(L.wm_conv2d_bwd_wb_partials_f32(
            x.data_ptr(), gy.data_ptr(), ws.data_ptr(), ws.numel(), N, Cin, H, W, Cout, kH, kW,
            ctypes.cast(layout, ctypes.c_void_p), _stream()))
    return GradPartials(ws, layout[0], layout[1], w_shape)


def adam_step_partials(params, partials, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                       grad_scale=1.0, grads_out=None):
    """Adam on the flat [w | b] parameter vector of one layer, with the fixed-order sum of the
    deferred gradient's partials fused in (one launch instead of finalize + Adam; same bits)."""
    if not isinstance(partials, GradPartials):
        raise TypeError("adam_step_partials: partials must come from conv2d_backward_wb_partials")
    count = partials.nw + partials.nb
    for name, t in (("params", params), ("m", m), ("v", v), ("grads_out", grads_out)):
        if t is None and name == "grads_out":
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or \
                not t.is_contiguous():
            raise TypeError(f"adam_step_partials: {name} must be a contiguous float32 CUDA tensor")
        if t.numel() != count or t.device != partials.ws.device:
            raise ValueError(f"adam_step_partials: {name} has {t.numel()} elements on {t.device}, "
                             f"the layer has {count} parameters on {partials.ws.device}")
    if int(step) < 1:
        raise ValueError("adam_step_partials: step counts from 1")
    with torch.cuda.device(params.device):
        _lib.check(_lib.lib().wm_adam_step_partials_f32(
            params.data_ptr(), partials.ws.data_ptr(), partials.nparts, partials.entry_major,
            m.data_ptr(), v.data_ptr(), count, 0 if grads_out is None else grads_out.data_ptr(),
            int(step), float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
            _stream()))


class Conv2dFunction(torch.autograd.Function):
    """autograd node so ``loss.backward()`` at the reference call site (newConvolution.py:184)
    reaches the HIP backward kernels."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return conv2d_forward(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = conv2d_backward_data(gy, w, x.shape) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = conv2d_backward_wb(x, gy, w.shape, need_bias=ctx.has_bias)
        return gx, gw, gb


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """Drop-in for ``torch.nn.functional.conv2d`` restricted to the convention every reference
    call uses (stride 1, no padding, dilation 1, groups 1); accepts the unbatched (C,H,W) input
    the reference passes (newConvolution.py:149)."""
    def _one(v):
        return v if isinstance(v, int) else (v[0] if len(set(v)) == 1 else None)
    if _one(stride) != 1 or _one(padding) != 0 or _one(dilation) != 1 or groups != 1:
        raise NotImplementedError("weather_model_amd.conv2d implements stride=1, padding=0, "
                                  "dilation=1, groups=1 (the reference's convention) only")
    unbatched = input.dim() == 3
    x = input.unsqueeze(0) if unbatched else input
    y = Conv2dFunction.apply(x, weight, bias)
    return y.squeeze(0) if unbatched else y


# --------------------------------------------------------------------------- conv3d / pool

def _conv3_dims(x, w):
    if x.dim() != 5 or w.dim() != 5:
        raise ValueError(f"conv3d expects (B,Cin,D,H,W) input and (Cout,Cin,kD,kH,kW) weight, got "
                         f"{tuple(x.shape)} and {tuple(w.shape)}")
    B, Cin, D, H, W = x.shape
    Cout, Cin2, kD, kH, kW = w.shape
    if Cin != Cin2:
        raise RuntimeError(f"conv3d: expected input with {Cin2} channels, got {Cin}")
    if kD > D or kH > H or kW > W:
        raise RuntimeError(f"conv3d: kernel size ({kD},{kH},{kW}) can't be greater than input "
                           f"({D},{H},{W})")
    return B, Cin, D, H, W, Cout, kD, kH, kW


def conv3d_forward(x, w, b=None, relu=False):
    x, w = _req(x, "input"), _req(w, "weight")
    b = None if b is None else _req(b, "bias")
    dims = _conv3_dims(x, w)
    B, Cin, D, H, W, Cout, kD, kH, kW = dims
    y = torch.empty((B, Cout, D - kD + 1, H - kH + 1, W - kW + 1), device=x.device,
                    dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().wm_conv3d_fwd_f32(
            x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(), y.data_ptr(), *dims,
            int(bool(relu)), _stream()))
    return y


def conv3d_backward_data(gy, w, x_shape):
    gy, w = _req(gy, "grad_output"), _req(w, "weight")
    B, Cin, D, H, W = x_shape
    Cout, _, kD, kH, kW = w.shape
    gx = torch.empty((B, Cin, D, H, W), device=gy.device, dtype=torch.float32)
    with torch.cuda.device(gy.device):
        _lib.check(_lib.lib().wm_conv3d_bwd_data_f32(
            gy.data_ptr(), w.data_ptr(), gx.data_ptr(), B, Cin, D, H, W, Cout, kD, kH, kW,
            _stream()))
    return gx


def conv3d_backward_wb(x, gy, w_shape, need_bias=True):
    x, gy = _req(x, "input"), _req(gy, "grad_output")
    B, Cin, D, H, W = x.shape
    Cout, _, kD, kH, kW = w_shape
    dims = (B, Cin, D, H, W, Cout, kD, kH, kW)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = L.wm_conv3d_bwd_wb_workspace_bytes(*dims)
        ws = torch.empty((max(nbytes, 8),), device=x.device, dtype=torch.uint8)
        gw = torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32)
        gb = torch.empty((Cout,), device=x.device, dtype=torch.float32) if need_bias else None
        _lib.check(L.wm_conv3d_bwd_wb_f32(
            x.data_ptr(), gy.data_ptr(), gw.data_ptr(), 0 if gb is None else gb.data_ptr(),
            ws.data_ptr(), ws.numel(), *dims, _stream()))
    return gw, gb


class Conv3dFunction(torch.autograd.Function):
    """autograd node for the Conv3D layers of convolution.py:164-172."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return conv3d_forward(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = conv3d_backward_data(gy, w, x.shape) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = conv3d_backward_wb(x, gy, w.shape, need_bias=ctx.has_bias)
        return gx, gw, gb


def conv3d(input, weight, bias=None):
    """3-D cross-correlation, valid padding, stride 1, channels_first."""
    return Conv3dFunction.apply(input, weight, bias)


def maxpool3d_forward(x, channels_last=True, want_index=True):
    """2x2x2 / stride-2 max pool of a 5-D tensor over axes (1,2,3) (``channels_last``, the Keras
    default the reference runs with) or (2,3,4).  Returns (y, which)."""
    x = _req(x, "input")
    if x.dim() != 5:
        raise ValueError(f"maxpool3d expects a 5-D tensor, got {tuple(x.shape)}")
    if channels_last:
        B, A1, A2, A3, C = x.shape
        out_shape = (B, A1 // 2, A2 // 2, A3 // 2, C)
    else:
        b0, c0, A1, A2, A3 = x.shape
        B, C = b0 * c0, 1
        out_shape = (b0, c0, A1 // 2, A2 // 2, A3 // 2)
    if min(A1, A2, A3) < 2:
        raise RuntimeError(f"maxpool3d: pooled axes ({A1},{A2},{A3}) are smaller than the window")
    y = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    which = torch.empty(out_shape, device=x.device, dtype=torch.uint8) if want_index else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().wm_maxpool3d_fwd_f32(
            x.data_ptr(), y.data_ptr(), 0 if which is None else which.data_ptr(), B, A1, A2, A3, C,
            _stream()))
    return y, which


def maxpool3d_backward(gy, which, x_shape, channels_last=True):
    gy = _req(gy, "grad_output")
    which = _req(which, "which", torch.uint8)
    if channels_last:
        B, A1, A2, A3, C = x_shape
    else:
        b0, c0, A1, A2, A3 = x_shape
        B, C = b0 * c0, 1
    gx = torch.empty(tuple(x_shape), device=gy.device, dtype=torch.float32)
    with torch.cuda.device(gy.device):
        _lib.check(_lib.lib().wm_maxpool3d_bwd_f32(gy.data_ptr(), which.data_ptr(), gx.data_ptr(),
                                                   B, A1, A2, A3, C, _stream()))
    return gx


class MaxPool3dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, channels_last):
        y, which = maxpool3d_forward(x, channels_last)
        ctx.save_for_backward(which)
        ctx.x_shape, ctx.channels_last = tuple(x.shape), channels_last
        return y

    @staticmethod
    def backward(ctx, gy):
        (which,) = ctx.saved_tensors
        return maxpool3d_backward(gy.contiguous(), which, ctx.x_shape, ctx.channels_last), None


def maxpool3d(input, channels_last=True):
    return MaxPool3dFunction.apply(input, channels_last)


class MaxPoolCatFunction(torch.autograd.Function):
    """``torch.cat((pool(xa).flatten(1), pool(xb).flatten(1)), 1)`` for two tensors of one shape --
    the end of the reference's two conv branches (convolution.py:164-174) -- as ONE launch that writes
    the concatenated matrix directly (wm_maxpool3d_fwd_pair_f32); the backward reads the two halves of
    the incoming gradient in place (wm_maxpool3d_bwd_pair_f32): no concatenation or slice copies."""

    @staticmethod
    def forward(ctx, xa, xb):
        xa, xb = _req(xa, "input a"), _req(xb, "input b")
        if xa.dim() != 5 or xa.shape != xb.shape:
            raise ValueError(f"maxpool3d_cat expects two 5-D tensors of one shape, got "
                             f"{tuple(xa.shape)} and {tuple(xb.shape)}")
        B, A1, A2, A3, C = xa.shape
        if min(A1, A2, A3) < 2:
            raise RuntimeError(f"maxpool3d: pooled axes ({A1},{A2},{A3}) are smaller than the window")
        per = (A1 // 2) * (A2 // 2) * (A3 // 2) * C
        h = torch.empty((B, 2 * per), device=xa.device, dtype=torch.float32)
        which = torch.empty((2, B, per), device=xa.device, dtype=torch.uint8)
        with torch.cuda.device(xa.device):
            _lib.check(_lib.lib().wm_maxpool3d_fwd_pair_f32(
                xa.data_ptr(), xb.data_ptr(), h.data_ptr(), h.data_ptr() + 4 * per,
                which[0].data_ptr(), which[1].data_ptr(), B, A1, A2, A3, C, 2 * per, _stream()))
        ctx.save_for_backward(which)
        ctx.x_shape = tuple(xa.shape)
        return h

    @staticmethod
    def backward(ctx, gh):
        (which,) = ctx.saved_tensors
        B, A1, A2, A3, C = ctx.x_shape
        per = which.shape[2]
        gh = _req(gh.contiguous(), "grad_output")
        gxa = torch.empty(ctx.x_shape, device=gh.device, dtype=torch.float32)
        gxb = torch.empty(ctx.x_shape, device=gh.device, dtype=torch.float32)
        with torch.cuda.device(gh.device):
            _lib.check(_lib.lib().wm_maxpool3d_bwd_pair_f32(
                gh.data_ptr(), gh.data_ptr() + 4 * per, which[0].data_ptr(), which[1].data_ptr(),
                gxa.data_ptr(), gxb.data_ptr(), B, A1, A2, A3, C, 2 * per, _stream()))
        return gxa, gxb


def maxpool3d_cat(xa, xb):
    """(B, 2 * flat): the channels-last 2x2x2 pools of ``xa`` and ``xb``, flattened, side by side."""
    return MaxPoolCatFunction.apply(xa, xb)


def copy_multi(dsts, srcs):
    """``dst[k].copy_(src[k])`` for contiguous CUDA tensors of equal size and dtype, all in one launch
    (wm_copy_multi; graph-capturable)."""
    if len(dsts) != len(srcs) or not dsts:
        raise ValueError("copy_multi: need as many sources as destinations, at least one")
    rec = np.zeros((len(dsts), 3), dtype=np.int64)
    for k, (d, s_) in enumerate(zip(dsts, srcs)):
        if not (d.is_cuda and s_.is_cuda and d.device == s_.device and d.dtype == s_.dtype and
                d.shape == s_.shape and d.is_contiguous() and s_.is_contiguous()):
            raise ValueError(f"copy_multi: pair {k} must be contiguous CUDA tensors of one shape and "
                             f"dtype on one device")
        rec[k] = (d.data_ptr(), s_.data_ptr(), d.numel() * d.element_size())
    with torch.cuda.device(dsts[0].device):
        _lib.check(_lib.lib().wm_copy_multi(rec.ctypes.data, len(dsts), _stream()))


# --------------------------------------------------------------------------- least squares

def gram(X, Y, acc=None):
    """acc (D x D float64, D = F+1+P) += [X|1|Y]' [X|1|Y].  acc is created zeroed if None."""
    X = _req(X, "X")
    Y = _req(Y, "Y")
    if X.dim() != 2:
        raise ValueError(f"X must be (n, F), got {tuple(X.shape)}")
    if Y.dim() == 1:
        Y = Y[:, None]
    if Y.dim() != 2 or Y.shape[0] != X.shape[0]:
        raise ValueError(f"Y must be (n, P) with n={X.shape[0]}, got {tuple(Y.shape)}")
    Y = Y.contiguous()
    n, F = X.shape
    P = Y.shape[1]
    if F < 1 or P < 1:
        raise ValueError("gram needs at least one feature and one target column")
    D = F + 1 + P
    if acc is None:
        acc = torch.zeros((D, D), device=X.device, dtype=torch.float64)
    else:
        if acc.shape != (D, D) or acc.dtype != torch.float64 or not acc.is_cuda or \
                not acc.is_contiguous():
            raise ValueError(f"acc must be a contiguous CUDA float64 ({D},{D}) tensor")
    if n == 0:
        return acc
    L = _lib.lib()
    with torch.cuda.device(X.device):
        nbytes = L.wm_gram_workspace_bytes(n, F, P)
        ws = torch.empty((max(nbytes, 8),), device=X.device, dtype=torch.uint8)
        _lib.check(L.wm_gram_f32(X.data_ptr(), Y.data_ptr(), n, F, P, acc.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _stream()))
    return acc


class GramPlan:
    """Pre-allocated workspace for repeated Gram builds of one shape (no per-call allocation)."""

    def __init__(self, n, F, P, device):
        self.n, self.F, self.P = int(n), int(F), int(P)
        self.device = torch.device(device)
        L = _lib.lib()
        with torch.cuda.device(self.device):
            nbytes = L.wm_gram_workspace_bytes(self.n, self.F, self.P)
        self.ws = torch.empty((max(nbytes, 8),), device=self.device, dtype=torch.uint8)
        D = F + 1 + P
        self.acc = torch.zeros((D, D), device=self.device, dtype=torch.float64)
        self.beta = torch.empty((F + 1, P), device=self.device, dtype=torch.float64)
        self.status = torch.zeros((1,), device=self.device, dtype=torch.int32)
        sb = L.wm_normal_solve_workspace_bytes(self.F, self.P)
        self.solve_ws = torch.empty((max(sb, 8),), device=self.device, dtype=torch.uint8)

    def _rows(self, t, name, cols):
        """A (n, cols) float32 contiguous tensor on this plan's device with n <= the planned rows."""
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device != self.acc.device:
            raise RuntimeError(f"{name} must be a tensor on {self.acc.device} (no CPU fallback)")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError(f"{name} must be contiguous float32, got {t.dtype}, "
                            f"contiguous={t.is_contiguous()}")
        if t.dim() == 1 and cols == 1:
            t = t[:, None]
        if t.dim() != 2 or t.shape[1] != cols:
            raise ValueError(f"{name} must be (n, {cols}), got {tuple(t.shape)}")
        if t.shape[0] > self.n:
            raise ValueError(f"{name} has {t.shape[0]} rows, the plan was made for {self.n}")
        return t

    def accumulate(self, X, Y):
        """acc += Gram of the rows of X, Y (any row count up to the planned n)."""
        X = self._rows(X, "X", self.F)
        Y = self._rows(Y, "Y", self.P)
        if Y.shape[0] != X.shape[0]:
            raise ValueError(f"X has {X.shape[0]} rows, Y has {Y.shape[0]}")
        if X.shape[0] == 0:
            return
        L = _lib.lib()
        with torch.cuda.device(self.acc.device):
            _lib.check(L.wm_gram_f32(X.data_ptr(), Y.data_ptr(), X.shape[0], self.F, self.P,
                                     self.acc.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                     _stream()))

    def solve(self, fit_intercept=True, ridge=0.0):
        L = _lib.lib()
        with torch.cuda.device(self.acc.device):
            _lib.check(L.wm_normal_solve_f64(self.acc.data_ptr(), self.F, self.P,
                                             int(bool(fit_intercept)), float(ridge),
                                             self.beta.data_ptr(), self.status.data_ptr(),
                                             self.solve_ws.data_ptr(), self.solve_ws.numel(),
                                             _stream()))
        return self.beta


def normal_solve(acc, F, P, fit_intercept=True, ridge=0.0, check=True):
    """beta ((F+1) x P float64, intercept last) from the Gram ``acc``."""
    D = F + 1 + P
    if acc.shape != (D, D) or acc.dtype != torch.float64 or not acc.is_cuda:
        raise ValueError(f"acc must be a CUDA float64 ({D},{D}) tensor")
    acc = acc.contiguous()
    L = _lib.lib()
    beta = torch.empty((F + 1, P), device=acc.device, dtype=torch.float64)
    status = torch.zeros((1,), device=acc.device, dtype=torch.int32)
    with torch.cuda.device(acc.device):
        nbytes = L.wm_normal_solve_workspace_bytes(F, P)
        ws = torch.empty((max(nbytes, 8),), device=acc.device, dtype=torch.uint8)
        _lib.check(L.wm_normal_solve_f64(acc.data_ptr(), F, P, int(bool(fit_intercept)),
                                         float(ridge), beta.data_ptr(), status.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream()))
    if check and int(status.item()) != 0:
        raise RuntimeError("normal equations are not positive definite (collinear features or "
                           "no rows); pass ridge > 0")
    return beta


def lstsq_fit(X, Y, fit_intercept=True, ridge=0.0):
    Yv = Y if Y.dim() == 2 else Y[:, None]
    acc = gram(X, Yv)
    return normal_solve(acc, X.shape[1], Yv.shape[1], fit_intercept, ridge)


def _wide_ok(L, F, P, *tensors):
    """The MFMA route for wide linear maps: shape supported, every pointer 16-byte aligned, and not
    switched off (WM_LINMETRICS_WIDE=0: probe switch)."""
    import os
    return (os.environ.get("WM_LINMETRICS_WIDE", "1") != "0" and bool(L.wm_linear_metrics_wide_supported(F, P))
            and all(t.data_ptr() % 16 == 0 for t in tensors))


def linear_wide(X, beta, Y=None, sums=None, want_msle=True, msle_eps=1e-7, predict=False):
    """wm_linear_metrics_wide_f32: the MFMA route for wide linear maps.  Adds the residual sums
    against ``Y`` to ``sums`` (when given) and / or returns the float32 predictions (``predict``).
    Returns (sums, predictions or None)."""
    L = _lib.lib()
    n, F = X.shape
    P = beta.shape[1]
    out = torch.empty((n, P), device=X.device, dtype=torch.float32) if predict else None
    if n == 0:
        return sums, out
    with torch.cuda.device(X.device):
        ws = torch.empty((max(L.wm_linear_metrics_wide_workspace_bytes(F, P), 16),), device=X.device,
                         dtype=torch.uint8)
        _lib.check(L.wm_linear_metrics_wide_f32(
            X.data_ptr(), None if Y is None else Y.data_ptr(), beta.data_ptr(), n, F, P,
            int(bool(want_msle)), float(msle_eps), None if sums is None else sums.data_ptr(),
            None if out is None else out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return sums, out


def linear_predict(X, beta):
    """model.predict of the fitted linear map.  Narrow maps: one thread per output, float64 dot
    products (wm_linear_predict_f32).  Wide maps (F * P >= 4096, e.g. singleRegression's 432 -> 216):
    the MFMA kernel of wm_linear_metrics_wide_f32 writing its float32 accumulators."""
    X = _req(X, "X")
    beta = _req(beta, "beta", torch.float64)
    n, F = X.shape
    P = beta.shape[1]
    if beta.shape[0] != F + 1:
        raise ValueError(f"beta must be ({F + 1}, P), got {tuple(beta.shape)}")
    L = _lib.lib()
    if n > 0 and F * P >= 4096 and _wide_ok(L, F, P, X):
        return linear_wide(X, beta, predict=True)[1]
    out = torch.empty((n, P), device=X.device, dtype=torch.float32)
    with torch.cuda.device(X.device):
        _lib.check(L.wm_linear_predict_f32(X.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                           n, F, P, _stream()))
    return out


def linear_metrics(X, Y, beta, sums=None, want_msle=True, msle_eps=1e-7):
    """Add {sum d^2, sum |d|, sum (log1p diff)^2, count} of ``[X | 1] beta - Y`` to the float64[4]
    ``sums`` (created zeroed when None) in one pass over the rows -- the loss / mae / msle that
    model.fit reports per epoch (multipleRegression.py:292-296).  Wide maps (beyond
    wm_linear_metrics_supported; singleRegression's 432 -> 216) take the MFMA kernel
    wm_linear_metrics_wide_f32; shapes neither takes (P not a multiple of 4, P > 224) form the
    predictions with a library GEMM first."""
    X = _req(X, "X")
    beta = _req(beta, "beta", torch.float64)
    Yv = _req(Y if Y.dim() == 2 else Y[:, None], "Y")
    n, F = X.shape
    P = beta.shape[1]
    if beta.shape[0] != F + 1 or Yv.shape != (n, P):
        raise ValueError(f"beta must be ({F + 1}, P) and Y ({n}, P); got {tuple(beta.shape)}, "
                         f"{tuple(Yv.shape)}")
    if sums is None:
        sums = torch.zeros((4,), device=X.device, dtype=torch.float64)
    elif sums.shape != (4,) or sums.dtype != torch.float64 or not sums.is_cuda:
        raise ValueError("sums must be a CUDA float64 tensor of 4")
    L = _lib.lib()
    with torch.cuda.device(X.device):
        ws = torch.empty((max(L.wm_linear_metrics_workspace_bytes(), 8),), device=X.device,
                         dtype=torch.uint8)
        if L.wm_linear_metrics_supported(F, P):
            _lib.check(L.wm_linear_metrics_f32(X.data_ptr(), Yv.data_ptr(), beta.data_ptr(), n, F, P,
                                               int(bool(want_msle)), float(msle_eps),
                                               sums.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        elif _wide_ok(L, F, P, X, Yv):
            # the reference's singleRegression shape (432 -> 216): predictions on the MFMA, scored
            # from the accumulators, never written
            linear_wide(X, beta, Yv, sums, want_msle, msle_eps)
        else:
            b32 = beta.float()
            pred = torch.addmm(b32[F], X, b32[:F])
            _lib.check(L.wm_residual_sums_f32(pred.data_ptr(), Yv.data_ptr(), n * P,
                                              int(bool(want_msle)), float(msle_eps),
                                              sums.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return sums


# --------------------------------------------------------------------------- fields

def field_stats(x, out=None):
    """(min, max, sum, sumsq, count) of a float32 CUDA tensor as a float64 tensor of 5; passing the
    previous ``out`` merges (running statistics over files)."""
    x = _req(x, "field").reshape(-1)
    if x.numel() == 0:
        raise ValueError("field_stats of an empty field")
    accumulate = out is not None
    if out is None:
        out = torch.empty((5,), device=x.device, dtype=torch.float64)
    elif out.shape != (5,) or out.dtype != torch.float64 or not out.is_cuda:
        raise ValueError("out must be a CUDA float64 tensor of 5")
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = L.wm_field_stats_workspace_bytes(x.numel())
        ws = torch.empty((max(nbytes, 8),), device=x.device, dtype=torch.uint8)
        _lib.check(L.wm_field_stats_f32(x.data_ptr(), x.numel(), out.data_ptr(), int(accumulate),
                                        ws.data_ptr(), ws.numel(), _stream()))
    return out


def affine_scale(x, offset, divisor, out=None):
    """(x - offset) / divisor in float32 (normalize / standardize of the reference)."""
    x = _req(x, "field")
    if out is not None and (not isinstance(out, torch.Tensor) or not out.is_contiguous()):
        raise ValueError("out must be a contiguous tensor (a copy would swallow the result)")
    y = torch.empty_like(x) if out is None else _req(out, "out")
    if y.shape != x.shape or y.device != x.device:
        raise ValueError("out has a different shape or device")
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().wm_affine_scale_f32(x.data_ptr(), y.data_ptr(), x.numel(),
                                                  float(offset), float(divisor), _stream()))
    return y


def gather_triplets(vol, shape, indices=None, first=0, count=None, time_offset=0, d_mid=3,
                    d_end=6):
    """x (B,2,V), y (B,V) for samples of a resident (T,Z,Y,X,V) volume; ``indices`` (int64 CUDA
    tensor) or the contiguous range [first, first + count)."""
    vol = _req(vol, "volume")
    if vol.dim() != 5:
        raise ValueError(f"volume must be (T,Z,Y,X,V), got {tuple(vol.shape)}")
    T, Z, Y, X, V = vol.shape
    St, Sz, Sy, Sx = (int(v) for v in shape)
    if indices is not None:
        indices = _req(indices, "indices", torch.int64).reshape(-1)
        B = indices.numel()
    else:
        B = int(count)
    xout = torch.empty((B, 2, V), device=vol.device, dtype=torch.float32)
    yout = torch.empty((B, V), device=vol.device, dtype=torch.float32)
    if B == 0:
        return xout, yout
    with torch.cuda.device(vol.device):
        _lib.check(_lib.lib().wm_gather_triplets_f32(
            vol.data_ptr(), 0 if indices is None else indices.data_ptr(), int(first),
            xout.data_ptr(), yout.data_ptr(), B, V, T, Z, Y, X, St, Sz, Sy, Sx, int(time_offset),
            int(d_mid), int(d_end), _stream()))
    return xout, yout


def gather_cubes(vol, cube, d_mid, d_end, rows, origins=None, walk=None, separate=False,
                 origins_device=None):
    """Rows of the regression / Conv3D generators from a resident (files, nvars, Z, Y, X) volume
    (``wm_gather_cubes_f32``).  ``cube`` = (cz, cy, cx).  Either ``origins`` -- an int32 (rows, 4)
    tensor/array of (file, z0, y0, x0) -- or ``walk`` = dict(batch, length, index, region=(r0,r1,r2),
    dims=(dim_z, dim_y, dim_x)), the generator state at the first batch.
    ``vol`` may be any strided view with that logical shape (``wm_gather_cubes_strided_f32``): a
    volume stored point-major, ``stored.permute(3, 4, 0, 1, 2)`` of a (Z, Y, X, files, nvars)
    tensor, turns the 4-byte gathers of cube-1 rows into contiguous reads.
    ``origins_device``: the same origins already on the device (int32, contiguous) -- ``origins``
    is then only checked on the host and not uploaded again (a pageable upload per call stalls the
    host behind the stream's queued work).
    Returns (X (rows, 2p), Y (rows, p)), or (x1 (rows, p), x2 (rows, p), Y) with ``separate``."""
    if not isinstance(vol, torch.Tensor) or not vol.is_cuda or vol.dtype != torch.float32:
        raise RuntimeError("volume must be a float32 CUDA tensor (there is no CPU fallback)")
    if vol.dim() != 5:
        raise ValueError(f"volume must be (files, nvars, Z, Y, X), got {tuple(vol.shape)}")
    if min(vol.stride()) < 0:
        raise ValueError("volume views with negative strides are not supported")
    strides = (ctypes.c_int64 * 5)(*vol.stride())
    files, nvars, Z, Y, X = vol.shape
    cz, cy, cx = (int(v) for v in cube)
    p = nvars * cz * cy * cx
    rows = int(rows)
    dev = vol.device
    if separate:
        x1 = torch.empty((rows, p), device=dev, dtype=torch.float32)
        x2 = torch.empty((rows, p), device=dev, dtype=torch.float32)
        a, b, stride = x1, x2, p
    else:
        Xo = torch.empty((rows, 2 * p), device=dev, dtype=torch.float32)
        a, b, stride = Xo, Xo[:, p:], 2 * p
    Yo = torch.empty((rows, p), device=dev, dtype=torch.float32)
    if rows > 0:
        dmax = max(int(d_mid), int(d_end))
        if origins is not None:
            oh = origins.cpu().numpy() if isinstance(origins, torch.Tensor) else np.asarray(origins)
            if oh.shape != (rows, 4):
                raise ValueError(f"origins must be (rows, 4), got {tuple(oh.shape)}")
            lo = oh.min(axis=0).tolist()
            hi = oh.max(axis=0).tolist()
            if min(lo) < 0 or hi[0] + dmax >= files or hi[1] + cz > Z or hi[2] + cy > Y or \
                    hi[3] + cx > X:
                raise ValueError("a cube origin leaves the volume")
            if origins_device is not None:
                o = origins_device
                if o.device != dev or o.dtype != torch.int32 or tuple(o.shape) != (rows, 4) or \
                        not o.is_contiguous():
                    raise ValueError("origins_device must be a contiguous int32 (rows, 4) tensor on "
                                     "the volume's device")
            else:
                o = torch.from_numpy(np.ascontiguousarray(oh, dtype=np.int32)).to(dev)
            args = (o.data_ptr(), rows, 1, 1, 0, 0, 0, 0, 1, 1, 1)
        else:
            r0, r1, r2 = (int(v) for v in walk["region"])
            dz, dy, dx = (int(v) for v in walk["dims"])
            args = (0, rows, int(walk["batch"]), int(walk["length"]), int(walk["index"]), r0, r1, r2,
                    dz, dy, dx)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().wm_gather_cubes_strided_f32(
                vol.data_ptr(), files, nvars, Z, Y, X, strides, cz, cy, cx, int(d_mid), int(d_end),
                *args,
                a.data_ptr(), b.data_ptr(), stride, Yo.data_ptr(), _stream()))
    return (x1, x2, Yo) if separate else (Xo, Yo)


def walk_fused_supported(vol, cube):
    """True when ``gram_walk`` / ``linear_metrics_walk`` take this volume: CUBE_SIZE 1 (the
    reference's setting, multipleRegression.py:35) and 1..4 variables."""
    return (isinstance(vol, torch.Tensor) and vol.is_cuda and vol.dtype == torch.float32 and
            vol.dim() == 5 and 1 <= vol.shape[1] <= 4 and tuple(int(c) for c in cube) == (1, 1, 1)
            and min(vol.stride()) >= 0)


def _walk_args(vol, d_mid, d_end, rows, walk):
    if not isinstance(vol, torch.Tensor) or not vol.is_cuda or vol.dtype != torch.float32:
        raise RuntimeError("volume must be a float32 CUDA tensor (there is no CPU fallback)")
    if vol.dim() != 5:
        raise ValueError(f"volume must be (files, nvars, Z, Y, X), got {tuple(vol.shape)}")
    if min(vol.stride()) < 0:
        raise ValueError("volume views with negative strides are not supported")
    files, nvars, Z, Y, X = vol.shape
    r0, r1, r2 = (int(v) for v in walk["region"])
    dz, dy, dx = (int(v) for v in walk["dims"])
    strides = (ctypes.c_int64 * 5)(*vol.stride())
    return nvars, (vol.data_ptr(), files, nvars, Z, Y, X, strides, int(d_mid), int(d_end), int(rows),
                   int(walk["batch"]), int(walk["length"]), int(walk["index"]), r0, r1, r2, dz, dy, dx)


def gram_walk(vol, d_mid, d_end, rows, walk, acc=None):
    """``gram(*gather_cubes(vol, (1, 1, 1), d_mid, d_end, rows, walk=walk), acc)`` without the rows:
    the Gram of the region walk's rows [x(t) | x(t + d_end)] ; x(t + d_mid) read straight from the
    resident volume (``wm_gram_walk_f32``; multipleRegression.py:217-270).  1..4 variables."""
    nvars, args = _walk_args(vol, d_mid, d_end, rows, walk)
    D = 3 * nvars + 1
    if acc is None:
        acc = torch.zeros((D, D), device=vol.device, dtype=torch.float64)
    elif acc.shape != (D, D) or acc.dtype != torch.float64 or not acc.is_cuda or \
            not acc.is_contiguous():
        raise ValueError(f"acc must be a contiguous CUDA float64 ({D},{D}) tensor")
    L = _lib.lib()
    with torch.cuda.device(vol.device):
        ws = torch.empty((max(L.wm_gram_walk_workspace_bytes(nvars), 8),), device=vol.device,
                         dtype=torch.uint8)
        _lib.check(L.wm_gram_walk_f32(*args, acc.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return acc


def linear_metrics_walk(vol, d_mid, d_end, rows, walk, beta, sums=None, want_msle=True,
                        msle_eps=1e-7):
    """``linear_metrics(*gather_cubes(...), beta, sums)`` without the rows
    (``wm_linear_metrics_walk_f32``)."""
    nvars, args = _walk_args(vol, d_mid, d_end, rows, walk)
    beta = _req(beta, "beta", torch.float64)
    if tuple(beta.shape) != (2 * nvars + 1, nvars):
        raise ValueError(f"beta must be ({2 * nvars + 1}, {nvars}), got {tuple(beta.shape)}")
    if sums is None:
        sums = torch.zeros((4,), device=vol.device, dtype=torch.float64)
    elif sums.shape != (4,) or sums.dtype != torch.float64 or not sums.is_cuda:
        raise ValueError("sums must be a CUDA float64 tensor of 4")
    L = _lib.lib()
    with torch.cuda.device(vol.device):
        ws = torch.empty((max(L.wm_linear_metrics_workspace_bytes(), 8),), device=vol.device,
                         dtype=torch.uint8)
        _lib.check(L.wm_linear_metrics_walk_f32(*args, beta.data_ptr(), int(bool(want_msle)),
                                                float(msle_eps), sums.data_ptr(), ws.data_ptr(),
                                                ws.numel(), _stream()))
    return sums


def dense_supported(in_features, out_features):
    """Shapes ``wm_dense_*`` take (16-byte rows)."""
    return in_features % 4 == 0 and out_features % 4 == 0


def dense_forward(x, weight, bias=None, relu=False):
    """act(x @ weight.T + bias) for x (M, K), weight (N, K) (``wm_dense_fwd_f32``)."""
    x, weight = _req(x, "input"), _req(weight, "weight")
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise ValueError(f"dense expects (M, K) and (N, K), got {tuple(x.shape)}, {tuple(weight.shape)}")
    M, K = x.shape
    N = weight.shape[0]
    b = None if bias is None else _req(bias, "bias")
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    if M == 0:
        return y
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().wm_dense_fwd_f32(x.data_ptr(), weight.data_ptr(),
                                               0 if b is None else b.data_ptr(), y.data_ptr(), M, N,
                                               K, 1 if relu else 0, _stream()))
    return y


def dense_backward(x, weight, gy, mask_below=None, need_gx=True, need_gw=True):
    """(gx, gW, gb) of ``dense_forward`` given the gradient gy of its PRE-activation output;
    ``mask_below`` (the post-ReLU activation that is this layer's input) zeroes gx where it is
    not positive (``wm_dense_bwd_f32``)."""
    x, weight, gy = _req(x, "input"), _req(weight, "weight"), _req(gy, "grad_output")
    M, K = x.shape
    N = weight.shape[0]
    L = _lib.lib()
    gx = torch.empty((M, K), device=x.device, dtype=torch.float32) if need_gx else None
    gw = torch.empty((N, K), device=x.device, dtype=torch.float32) if need_gw else None
    gb = torch.empty((N,), device=x.device, dtype=torch.float32) if need_gw else None
    mb = None if mask_below is None else _req(mask_below, "mask")
    with torch.cuda.device(x.device):
        ws = torch.empty((max(L.wm_dense_bwd_workspace_bytes(M, N, K), 16),), device=x.device,
                         dtype=torch.uint8)
        _lib.check(L.wm_dense_bwd_f32(
            x.data_ptr(), weight.data_ptr(), gy.data_ptr(), 0 if mb is None else mb.data_ptr(),
            0 if gx is None else gx.data_ptr(), 0 if gw is None else gw.data_ptr(),
            0 if gb is None else gb.data_ptr(), ws.data_ptr(), ws.numel(), M, N, K, _stream()))
    return gx, gw, gb


class DenseFunction(torch.autograd.Function):
    """y = act(x W^T + b).  ``relu_input``: the input x is a post-ReLU activation and this layer
    applies that ReLU's backward in its gx store.  ``relu_bwd_fused``: this layer's own ReLU
    backward is applied by its consumer (a DenseFunction with ``relu_input``), so the incoming
    gradient is already that of the pre-activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, relu_input, relu_bwd_fused):
        y = dense_forward(x, weight, bias, relu)
        own_mask = relu and not relu_bwd_fused
        ctx.save_for_backward(x, weight, y if own_mask else None)
        ctx.own_mask, ctx.relu_input, ctx.has_bias = own_mask, relu_input, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.own_mask:
            gy = relu_mask(gy, y)                     # stand-alone mask (no fusing consumer)
        gx, gw, gb = dense_backward(x, weight, gy, x if ctx.relu_input else None,
                                    need_gx=ctx.needs_input_grad[0],
                                    need_gw=ctx.needs_input_grad[1])
        return gx, gw, (gb if ctx.has_bias else None), None, None, None


def dense(x, weight, bias=None, relu=False, relu_input=False, relu_bwd_fused=False):
    return DenseFunction.apply(x, weight, bias, relu, relu_input, relu_bwd_fused)


def mse_metrics(pred, y, broadcast, eps=1e-7, want_grad=True):
    """Mean squared error, mean absolute error, mean squared logarithmic error and d(mse)/d(pred)
    of predictions (B, P) against targets (B, P) in one pass (``wm_mse_metrics_f32``); with
    ``broadcast`` every prediction is scored against every target of the batch (what the
    reference's (B, 1, P) targets do, convolution.py:137,214).  Returns (out3, grad): a float32
    device tensor [mse, mae, msle] and the gradient (or None).  B <= 4096."""
    pred = _req(pred, "pred")
    y = _req(y, "targets")
    if pred.dim() != 2 or y.shape != pred.shape:
        raise ValueError(f"pred and targets must both be (B, P), got {tuple(pred.shape)} and "
                         f"{tuple(y.shape)}")
    B, P = pred.shape
    L = _lib.lib()
    with torch.cuda.device(pred.device):
        ws = torch.empty((max(L.wm_mse_metrics_workspace_bytes(P), 8),), device=pred.device,
                         dtype=torch.uint8)
        out3 = torch.empty((3,), device=pred.device, dtype=torch.float32)
        grad = torch.empty_like(pred) if want_grad else None
        _lib.check(L.wm_mse_metrics_f32(pred.data_ptr(), y.data_ptr(), B, P, 1 if broadcast else 0,
                                        float(eps), 0 if grad is None else grad.data_ptr(),
                                        out3.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out3, grad


# --------------------------------------------------------------------------- synthetic data

def synth_uniform(shape, seed, lo=0.0, hi=1.0, offset=0, device="cuda"):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().wm_synth_uniform_f32(out.data_ptr(), out.numel(), int(seed),
                                                   int(offset), float(lo), float(hi), _stream()))
    return out


SYNTH_DIST = {"uniform": 0, "normal": 1, "uniform_paired": 2, "uniform32": 3}


def synth_regression(n, F, P, seed, beta, intercept=0.5, noise=0.01, dist="uniform",
                     row_offset=0, device="cuda"):
    beta = _req(beta.to(device), "beta")
    X = torch.empty((n, F), device=device, dtype=torch.float32)
    Y = torch.empty((n, P), device=device, dtype=torch.float32)
    with torch.cuda.device(X.device):
        _lib.check(_lib.lib().wm_synth_regression_f32(
            X.data_ptr(), Y.data_ptr(), n, F, P, int(seed), int(row_offset), beta.data_ptr(),
            float(intercept), float(noise), SYNTH_DIST[dist], _stream()))
    return X, Y


# --------------------------------------------------------------------------- fused model

MODEL_PARAM_SHAPES = (
    ("conv1.weight", (16, 1, 2, 1)), ("conv1.bias", (16,)),
    ("conv2.weight", (32, 16, 3, 1)), ("conv2.bias", (32,)),
    ("conv3.weight", (32, 32, 3, 1)), ("conv3.bias", (32,)),
    ("conv4.weight", (32, 32, 3, 1)), ("conv4.bias", (32,)),
    ("linear1.weight", (64, 192)), ("linear1.bias", (64,)),
    ("linear2.weight", (6, 64)), ("linear2.bias", (6,)),
)
MODEL_PARAM_COUNT = 20566


def _model_x(x):
    """(B,2,6), (B,1,2,6) or the reference's single (1,2,6)/(2,6) sample -> contiguous (B,2,6)."""
    x = _req(x, "x")
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if x.dim() == 4 and x.shape[1] == 1:
        x = x.squeeze(1)
    if x.dim() != 3 or x.shape[1:] != (2, 6):
        raise ValueError(f"model input must be (B,2,6) or (B,1,2,6), got {tuple(x.shape)}")
    return x.contiguous()


def model_forward(flat_params, x):
    p = _req(flat_params, "params")
    if p.numel() != MODEL_PARAM_COUNT:
        raise ValueError(f"params must hold {MODEL_PARAM_COUNT} floats, got {p.numel()}")
    x = _model_x(x)
    B = x.shape[0]
    out = torch.empty((B, 6), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().wm_model_fwd_f32(p.data_ptr(), x.data_ptr(), out.data_ptr(), B,
                                               _stream()))
    return out


def model_train(flat_params, x, y=None, dout=None, batch_global=None, ws=None):
    """forward + backward.  Returns (out (B,6), loss (0-dim tensor or None), grads (20566,))."""
    p = _req(flat_params, "params")
    x = _model_x(x)
    B = x.shape[0]
    if (y is None) == (dout is None):
        raise ValueError("pass exactly one of y and dout")
    t = _req(y if y is not None else dout, "y/dout").reshape(B, 6).contiguous()
    L = _lib.lib()
    out = torch.empty((B, 6), device=x.device, dtype=torch.float32)
    grads = torch.empty((MODEL_PARAM_COUNT,), device=x.device, dtype=torch.float32)
    loss = torch.zeros((), device=x.device, dtype=torch.float32) if y is not None else None
    with torch.cuda.device(x.device):
        if ws is None:
            ws = torch.empty((L.wm_model_workspace_bytes(B),), device=x.device, dtype=torch.uint8)
        _lib.check(L.wm_model_train_f32(
            p.data_ptr(), x.data_ptr(), t.data_ptr() if y is not None else 0,
            t.data_ptr() if dout is not None else 0, out.data_ptr(),
            loss.data_ptr() if loss is not None else 0, grads.data_ptr(), ws.data_ptr(),
            ws.numel(), B, int(batch_global or B), _stream()))
    return out, loss, grads


def model_train_gather(flat_params, vol, shape, indices=None, first=0, count=None, time_offset=0,
                       d_mid=3, d_end=6, batch_global=None, ws=None):
    """``model_train(flat_params, *gather_triplets(vol, shape, ...))`` in one launch: the kernel
    reads its samples from the resident (T,Z,Y,X,6) volume by index (wm_model_train_gather_f32).
    Returns (out, loss, grads)."""
    p = _req(flat_params, "params")
    if p.numel() != MODEL_PARAM_COUNT:
        raise ValueError(f"params must hold {MODEL_PARAM_COUNT} floats, got {p.numel()}")
    vol = _req(vol, "volume")
    if vol.dim() != 5 or vol.shape[4] != 6:
        raise ValueError(f"volume must be (T,Z,Y,X,6), got {tuple(vol.shape)}")
    T, Z, Y, X, _ = vol.shape
    St, Sz, Sy, Sx = (int(v) for v in shape)
    if indices is not None:
        indices = _req(indices, "indices", torch.int64).reshape(-1)
        B = indices.numel()
    else:
        B = int(count)
    if B <= 0:
        raise ValueError("model_train_gather needs at least one sample")
    dev = vol.device
    out = torch.empty((B, 6), device=dev, dtype=torch.float32)
    grads = torch.empty((MODEL_PARAM_COUNT,), device=dev, dtype=torch.float32)
    loss = torch.zeros((), device=dev, dtype=torch.float32)
    L = _lib.lib()
    with torch.cuda.device(dev):
        if ws is None:
            ws = torch.empty((L.wm_model_workspace_bytes(B),), device=dev, dtype=torch.uint8)
        _lib.check(L.wm_model_train_gather_f32(
            p.data_ptr(), vol.data_ptr(), 0 if indices is None else indices.data_ptr(), int(first),
            T, Z, Y, X, St, Sz, Sy, Sx, int(time_offset), int(d_mid), int(d_end), out.data_ptr(),
            loss.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), B,
            int(batch_global or B), _stream()))
    return out, loss, grads


def adam_step(params, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """In-place torch.optim.Adam step on flat float32 buffers (newConvolution.py:185,225)."""
    for name, t in (("params", params), ("grads", grads), ("m", m), ("v", v)):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or \
                not t.is_contiguous():
            raise TypeError(f"adam_step: {name} must be a contiguous float32 CUDA tensor")
        if t.numel() != params.numel() or t.device != params.device:
            raise ValueError(f"adam_step: {name} has {t.numel()} elements on {t.device}, params has "
                             f"{params.numel()} on {params.device}")
    if int(step) < 1:
        raise ValueError("adam_step: step counts from 1")
    with torch.cuda.device(params.device):
        _lib.check(_lib.lib().wm_adam_step_f32(params.data_ptr(), grads.data_ptr(), m.data_ptr(),
                                               v.data_ptr(), params.numel(), int(step), float(lr),
                                               float(beta1), float(beta2), float(eps),
                                               float(grad_scale), _stream()))


class DeviceAdam(torch.optim.Optimizer):
    """Adam over a list of parameter tensors in ONE launch (wm_adam_multi_f32), step counter on the
    device -- capturable in a HIP graph.  ``keras_eps=True`` is keras.optimizers.Adam's update
    (epsilon outside the bias correction: what the reference's Conv3D model trains with,
    convolution.py:54,185); False is torch.optim.Adam's.  State per parameter: ``exp_avg``,
    ``exp_avg_sq`` and the shared device counter ``step``.

    ``overlap(params)`` moves the update of those tensors out of ``step()`` and into the backward
    pass: as soon as autograd has accumulated such a tensor's gradient its update is launched on a
    side stream, beside the rest of the backward pass (the Conv3D model's output layer holds 90 % of
    the weights and its gradient is the first one ready).  Only for parameters that nothing LATER in
    the backward pass reads."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-7, keras_eps=True):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.keras_eps = bool(keras_eps)
        self._step = None                  # device int32: updates made (moved on by a step's first launch)
        self._advanced = False             # this step's first launch has gone out
        self._tables = {}
        self._early = {}                   # id(param) -> hook handle
        self._early_done = []
        self._early_stream = None

    def _params(self):
        return [p for grp in self.param_groups for p in grp["params"]]

    def overlap(self, params):
        params = list(params)
        mine = {id(p) for p in self._params()}
        if any(id(p) not in mine for p in params):
            raise ValueError("DeviceAdam.overlap: not a parameter of this optimizer")
        if len({id(p) for p in params} | set(self._early)) >= len(mine):
            raise ValueError("DeviceAdam.overlap: at least one parameter must stay in step()")
        for p in params:
            if id(p) not in self._early:
                self._early[id(p)] = p.register_post_accumulate_grad_hook(self._early_hook)

    @torch.no_grad()
    def _early_hook(self, p):
        if p.grad is None:
            return
        main = torch.cuda.current_stream(p.device)
        side = self._early_stream
        if side is None or side.device != p.device:
            side = self._early_stream = torch.cuda.Stream(p.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._apply([p])
        if not torch.cuda.is_current_stream_capturing():
            p.grad.record_stream(side)
        self._early_done.append(p)

    def _apply(self, ps):
        dev = ps[0].device
        advance, self._advanced = not self._advanced, True
        if self._step is None:
            self._step = torch.zeros((), dtype=torch.int32, device=dev)
        for p in ps:
            st = self.state[p]
            if not st:
                st["step"] = self._step
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            for name, t in (("param", p), ("grad", p.grad)):
                if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError(f"DeviceAdam: {name} must be a contiguous float32 CUDA tensor")
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                     self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in ps)
        ent = self._tables.get(key)
        if ent is None:
            rec = np.zeros((len(ps), 6), dtype=np.int64)
            blocks = 0
            for k, row in enumerate(key):
                rec[k, :5] = row
                rec[k, 5] = blocks
                blocks += (row[4] + 1023) // 1024
            ent = self._tables[key] = (rec, blocks)       # host records: they go in as launch arguments
            if len(self._tables) > 64:
                self._tables = {key: ent}
        table, nblocks = ent
        grp = self.param_groups[0]
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().wm_adam_multi_f32(
                table.ctypes.data, len(ps), nblocks, self._step.data_ptr(), float(grp["lr"]),
                float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                1 if self.keras_eps else 0, 1 if advance else 0, _stream()))

    @torch.no_grad()
    def step(self, closure=None):
        done = {id(p) for p in self._early_done}
        ps = [p for p in self._params() if p.grad is not None and id(p) not in done]
        if not ps:
            if done:
                raise RuntimeError("DeviceAdam: every gradient of this step went through overlap()")
            return None
        if done:                            # the step ends when the early updates have
            torch.cuda.current_stream(ps[0].device).wait_stream(self._early_stream)
            self._early_done = []
        self._apply(ps)
        self._advanced = False
        return None


class ModelFunction(torch.autograd.Function):
    """out = CustomModel(x) with the fused kernels; backward receives dL/dout from autograd
    (e.g. from nn.MSELoss, newConvolution.py:181-184) and returns the flat parameter gradient."""

    @staticmethod
    def forward(ctx, flat_params, x):
        ctx.save_for_backward(flat_params, x)
        return model_forward(flat_params, x)

    @staticmethod
    def backward(ctx, gout):
        flat_params, x = ctx.saved_tensors
        _, _, grads = model_train(flat_params, x, dout=gout.contiguous())
        return grads, None
