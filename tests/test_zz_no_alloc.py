"""GPU: the allocation-free entry point wm_conv2d_bwd_wb_noalloc_f32 (static device scratch in place
of the caller workspace) against wm_conv2d_bwd_wb_f32, through the C ABI.  Integer/index bar does not
apply (float sums), but both entry points run the same kernels in the same fixed order, so the bar
here is BITWISE equality; the existing entry point is itself checked against the float64 oracle at
1e-5 in test_gpu_conv.py.  Two calls in a row, the second with a smaller input: the scratch keeps the
first call's partial sums and the second call must not read them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, check, x, gy, wshape):
    Cout, Cin, kH, kW = wshape
    N, _, H, W = x.shape
    st = torch.cuda.current_stream().cuda_stream
    nent = Cout * Cin * kH * kW
    need = L.wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW)
    ws = torch.empty((need,), device="cuda", dtype=torch.uint8)
    gw0 = torch.full((nent,), float("nan"), device="cuda")
    gb0 = torch.full((Cout,), float("nan"), device="cuda")
    gw1, gb1 = gw0.clone(), gb0.clone()
    check(L.wm_conv2d_bwd_wb_f32(x.data_ptr(), gy.data_ptr(), gw0.data_ptr(), gb0.data_ptr(),
                                 ws.data_ptr(), need, N, Cin, H, W, Cout, kH, kW, st))
    check(L.wm_conv2d_bwd_wb_noalloc_f32(x.data_ptr(), gy.data_ptr(), gw1.data_ptr(),
                                         gb1.data_ptr(), N, Cin, H, W, Cout, kH, kW, st))
    torch.cuda.synchronize()
    return (gw0, gb0), (gw1, gb1)


def test_noalloc_matches_workspace_entry_point_twice_second_smaller():
    from weather_model_amd import _lib, ops
    from oracle import cpu as ocpu
    from conftest import relinf
    L = _lib.lib()
    assert L.wm_conv2d_bwd_wb_noalloc_max_bytes() == 1 << 20
    shapes = [((6, 1, 181, 360), (1, 1, 3, 3)),      # flat stencil kernel, 390 k grid points
              ((2, 1, 37, 52), (1, 1, 3, 3)),        # smaller: fewer workgroups than the call before
              ((3, 1, 33, 130), (1, 1, 5, 5)),       # strip kernel
              ((2, 3, 8, 9), (2, 3, 2, 3))]          # generic kernel
    for k, (xs, wsh) in enumerate(shapes):
        N, Cin, H, W = xs
        Cout, _, kH, kW = wsh
        x = ops.synth_uniform(xs, seed=300 + k)
        gy = ops.synth_uniform((N, Cout, H - kH + 1, W - kW + 1), seed=400 + k, lo=-1.0, hi=1.0)
        (gw0, gb0), (gw1, gb1) = _both(L, _lib.check, x, gy, wsh)
        assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1), (xs, wsh)
        assert bool(torch.isfinite(gw1).all()) and bool(torch.isfinite(gb1).all())
        if k == 1:     # and against the oracle once (x ~ U(0,1), gy ~ U(-1,1))
            ogw, ogb = ocpu.conv2d_bwd_wb(x.cpu().numpy(), gy.cpu().numpy(), wsh)
            assert relinf(gw1.cpu().numpy().reshape(wsh), ogw) < 1e-5


def test_noalloc_rejects_shapes_beyond_its_scratch():
    from weather_model_amd import _lib
    L = _lib.lib()
    N, C, H, W = 1, 32, 64, 256              # implicit-GEMM path: float64 slabs of 9248 entries each
    need = L.wm_conv2d_bwd_wb_workspace_bytes(N, C, H, W, C, 3, 3)
    assert need > L.wm_conv2d_bwd_wb_noalloc_max_bytes()
    x = torch.zeros((N, C, H, W), device="cuda")
    gy = torch.zeros((N, C, H - 2, W - 2), device="cuda")
    gw = torch.zeros((C * C * 9,), device="cuda")
    rc = L.wm_conv2d_bwd_wb_noalloc_f32(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), 0,
                                        N, C, H, W, C, 3, 3, 0)
    assert rc == -2 and b"static scratch" in L.wm_last_error()
