"""GPU parity: HIP conv2d (through the C ABI) vs the float64 oracle and the golden fixtures.
Tolerance (BASELINE.json north_star): 1e-5 relative, ||a-b||_inf / ||b||_inf, float32."""
import numpy as np
import pytest
import torch

from conftest import relinf
from cases import CONV_CASES, conv_inputs, grid721_inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("ci", range(len(CONV_CASES)), ids=[c[0] for c in CONV_CASES])
def test_conv_matches_oracle_and_golden(ci, conv_golden):
    from weather_model_amd import ops
    from oracle import cpu as ocpu
    name = CONV_CASES[ci][0]
    x, w, b, gy = conv_inputs(ci)
    y = ops.conv2d_forward(_dev(x), _dev(w), _dev(b)).cpu().numpy()
    assert relinf(y, ocpu.conv2d_fwd(x, w, b)) < TOL
    assert relinf(y, conv_golden[f"{name}/y_f64"]) < TOL
    gx = ops.conv2d_backward_data(_dev(gy), _dev(w), x.shape).cpu().numpy()
    assert relinf(gx, conv_golden[f"{name}/gx_f64"]) < TOL
    gw, gb = ops.conv2d_backward_wb(_dev(x), _dev(gy), w.shape)
    assert relinf(gw.cpu().numpy(), conv_golden[f"{name}/gw_f64"]) < TOL
    assert relinf(gb.cpu().numpy(), conv_golden[f"{name}/gb_f64"]) < TOL
    # no-bias and fused-ReLU variants
    y0 = ops.conv2d_forward(_dev(x), _dev(w), None).cpu().numpy()
    assert relinf(y0, ocpu.conv2d_fwd(x, w, None)) < TOL
    yr = ops.conv2d_forward(_dev(x), _dev(w), _dev(b), relu=True).cpu().numpy()
    assert relinf(yr, np.maximum(ocpu.conv2d_fwd(x, w, b), 0.0)) < TOL


def test_conv_full_grid_721x1440(conv_golden):
    """BASELINE config 3 grid (2 images): checksums and corner tiles of the golden fixture."""
    from weather_model_amd import ops
    g = conv_golden
    x, w, b, gy = grid721_inputs()
    y = ops.conv2d_forward(_dev(x), _dev(w), _dev(b)).cpu().numpy().astype(np.float64)
    assert abs(y.sum() - g["grid721/y_sum"]) < 1e-6 * abs(g["grid721/y_sum"])
    assert abs((y ** 2).sum() - g["grid721/y_sumsq"]) < 1e-6 * g["grid721/y_sumsq"]
    assert relinf(y[:, :, :8, :8], g["grid721/y_tl"]) < TOL
    assert relinf(y[:, :, -8:, -8:], g["grid721/y_br"]) < TOL
    assert relinf(y[:, :, :8, -8:], g["grid721/y_tr"]) < TOL
    assert relinf(y[:, :, 355:363, 715:723], g["grid721/y_mid"]) < TOL
    gw, gb = ops.conv2d_backward_wb(_dev(x), _dev(gy), w.shape)
    assert relinf(gw.cpu().numpy(), g["grid721/gw"]) < TOL
    assert relinf(gb.cpu().numpy(), g["grid721/gb"]) < TOL
    gx = ops.conv2d_backward_data(_dev(gy), _dev(w), x.shape).cpu().numpy().astype(np.float64)
    assert abs((gx ** 2).sum() - g["grid721/gx_sumsq"]) < 1e-6 * g["grid721/gx_sumsq"]
    assert relinf(gx[:, :, :8, :8], g["grid721/gx_tl"]) < TOL
    assert relinf(gx[:, :, -8:, -8:], g["grid721/gx_br"]) < TOL


def test_conv3x3_full_size_adjoint_identities_and_batch_independence():
    """BASELINE config 3 at its full size, Conv2d(1 -> 1, 3x3) on (64,1,721,1440) (the flat stencil
    kernels):  <A x, g> = <x, A^T g> = <w, dW(x, g)>,  db = sum g,  and every image of the batch
    gives bit for bit what it gives alone (images are independent; the kernel's row ranges cross
    image boundaries)."""
    from weather_model_amd import ops
    N, H, W = 64, 721, 1440
    x = ops.synth_uniform((N, 1, H, W), seed=81)
    w = ops.synth_uniform((1, 1, 3, 3), seed=82, lo=-1.0 / 3, hi=1.0 / 3)
    b = ops.synth_uniform((1,), seed=83, lo=-1.0 / 3, hi=1.0 / 3)
    g = ops.synth_uniform((N, 1, H - 2, W - 2), seed=84, lo=-1.0, hi=1.0)

    def dot(a, c):
        return float((a.double().flatten() * c.double().flatten()).sum())

    y0 = ops.conv2d_forward(x, w, None)
    lhs = dot(y0, g)
    y = ops.conv2d_forward(x, w, b)
    assert float((y - y0 - b.view(1, 1, 1, 1)).abs().max()) < 1e-6
    for n in (0, 31, 63):
        assert torch.equal(ops.conv2d_forward(x[n:n + 1].contiguous(), w, b), y[n:n + 1])
    del y, y0
    gx = ops.conv2d_backward_data(g, w, x.shape)
    mid = dot(x, gx)
    del gx
    gw, gb = ops.conv2d_backward_wb(x, g, w.shape)
    rhs = dot(w, gw)
    scale = float((x.double() ** 2).sum().sqrt() * (g.double() ** 2).sum().sqrt()) / 3
    assert abs(lhs - mid) < 5e-8 * scale and abs(lhs - rhs) < 5e-8 * scale, (lhs, mid, rhs, scale)
    assert abs(float(gb.double()[0]) - float(g.double().sum())) < 1e-5 * float(g.double().abs().sum())


def test_conv32_full_size_adjoint_identities():
    """BASELINE config 4b at its full size, Conv2d(32 -> 32, 3x3) on (8,32,721,1440): the three
    passes checked against each other through the identities every linear map and its adjoints
    satisfy (no oracle needed at 1.2 TFLOP):  <A x, g> = <x, A^T g> = <w, dW(x, g)>  with A the
    bias-free convolution, and  db = sum g.  Dot products in float64 on the device."""
    import torch
    from weather_model_amd import ops
    N, C, H, W = 8, 32, 721, 1440
    x = ops.synth_uniform((N, C, H, W), seed=91)
    w = ops.synth_uniform((C, C, 3, 3), seed=92, lo=-1.0 / 17, hi=1.0 / 17)
    b = ops.synth_uniform((C,), seed=93, lo=-1.0, hi=1.0)
    g = ops.synth_uniform((N, C, H - 2, W - 2), seed=94, lo=-1.0, hi=1.0)

    def dot(a, c):
        return float((a.double().flatten() * c.double().flatten()).sum())

    y0 = ops.conv2d_forward(x, w, None)
    lhs = dot(y0, g)
    y = ops.conv2d_forward(x, w, b)
    # the bias enters once per output position
    assert torch.allclose(y[:, :, 100, 200] - y0[:, :, 100, 200], b.expand(N, C), atol=1e-6)
    del y, y0
    gx = ops.conv2d_backward_data(g, w, x.shape)
    mid = dot(x, gx)
    del gx
    gw, gb = ops.conv2d_backward_wb(x, g, w.shape)
    rhs = dot(w, gw)
    scale = float((x.double() ** 2).sum().sqrt() * (g.double() ** 2).sum().sqrt()) / 17
    assert abs(lhs - mid) < 5e-8 * scale and abs(lhs - rhs) < 5e-8 * scale, (lhs, mid, rhs, scale)
    gsum = g.double().sum((0, 2, 3))
    assert float((gb.double() - gsum).abs().max()) < 1e-5 * float(g.double().abs().sum((0, 2, 3)).max())
    # linearity of the weight gradient in gy (size-independent property)
    gw2, gb2 = ops.conv2d_backward_wb(x, 2.0 * g, w.shape)
    assert float((gw2 - 2.0 * gw).abs().max()) <= 2e-6 * float(gw.abs().max())


@pytest.mark.parametrize("shape", [
    (1, 1, 3, 19, 1, 3, 3),       # a single output row, OW = 17 (stencil path, ragged strip)
    (5, 1, 40, 66, 1, 3, 3),      # ranges that cross image boundaries
    (3, 1, 33, 130, 1, 5, 5),
    (2, 1, 9, 70, 1, 3, 1), (2, 1, 9, 70, 1, 1, 3), (2, 1, 9, 70, 1, 2, 2),
    (2, 1, 9, 70, 1, 1, 1), (4, 1, 2, 64, 1, 2, 1), (4, 1, 6, 65, 1, 1, 2),
    # W % 4 == 0: the flat-aligned kernels (LDS ring, shuffle halo, ragged head/tail of a range)
    (1, 1, 3, 20, 1, 3, 3),       # one output row, OW = 18: last lane stores 2 of 4
    (5, 1, 40, 68, 1, 3, 3),      # ranges that cross image boundaries
    (3, 1, 7, 256, 1, 3, 3),      # OW = 254: second wave holds two columns
    (2, 1, 9, 260, 1, 1, 3), (2, 1, 9, 72, 1, 3, 1), (2, 1, 9, 72, 1, 2, 2),
    (2, 1, 9, 72, 1, 1, 1), (4, 1, 6, 68, 1, 1, 2), (3, 1, 11, 24, 1, 2, 1),
    (7, 1, 3, 1024, 1, 3, 3),     # H == kH: one output row per image, every range crosses images
    (2, 1, 5, 4032, 1, 3, 3),     # 16 waves per row (the widest the flat kernels take)
    (1, 1, 4, 4100, 1, 3, 3),     # wider: falls back to the strip kernels
    (1, 1, 1500, 20, 1, 3, 3),    # tall and narrow: hundreds of short ranges
    (2, 16, 200, 400, 24, 3, 3),  # multi-channel, > 131072 outputs: wide MFMA kernel, ragged Cout
    (1, 20, 140, 1000, 40, 2, 3), # wide MFMA kernel, ragged Cin, three channel tiles
    (2, 16, 600, 900, 24, 3, 3),  # > 2^20 outputs: the persistent wide weight-gradient kernel
    # large multi-channel grids (row kernels; 3x3 with 17..32 output channels on the input-gradient
    # side also takes conv_ring.hip): ragged channel counts, kernels up to 5 x 4, the reference's
    # (3, 1) orientation, one 64-column stripe exactly, three channel tiles, one tile
    (4, 8, 140, 330, 24, 3, 3), (4, 6, 120, 300, 20, 5, 4), (6, 16, 200, 128, 32, 3, 1),
    (20, 32, 130, 66, 32, 3, 3), (5, 12, 100, 300, 48, 3, 3), (5, 20, 100, 300, 10, 2, 2),
    (3, 33, 150, 200, 35, 1, 2),
    (1, 1, 7, 7, 1, 7, 7),        # generic path, 1x1 output
    (2, 3, 8, 9, 2, 2, 3),        # generic path, odd everything
])
def test_conv_ragged_shapes_vs_oracle(shape):
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    N, Cin, H, W, Cout, kH, kW = shape
    x = synth.uniform(50, (N, Cin, H, W), -1, 1)
    w = synth.uniform(51, (Cout, Cin, kH, kW), -1, 1)
    b = synth.uniform(52, (Cout,), -1, 1)
    gy = synth.uniform(53, (N, Cout, H - kH + 1, W - kW + 1), -1, 1)
    y = ops.conv2d_forward(_dev(x), _dev(w), _dev(b)).cpu().numpy()
    assert relinf(y, ocpu.conv2d_fwd(x, w, b)) < TOL
    gx = ops.conv2d_backward_data(_dev(gy), _dev(w), x.shape).cpu().numpy()
    assert relinf(gx, ocpu.conv2d_bwd_data(gy, w, x.shape)) < TOL
    # weight/bias gradient: ||a-b||_inf / ||b||_inf against the float64 oracle, on inputs whose sums
    # do not cancel (x ~ U(0,1), gy ~ U(0,1)) so that the norm of the reference is the size of the
    # sums themselves -- no absolute floor
    xp = synth.uniform(54, (N, Cin, H, W), 0, 1)
    gp = synth.uniform(55, (N, Cout, H - kH + 1, W - kW + 1), 0, 1)
    gw, gb = ops.conv2d_backward_wb(_dev(xp), _dev(gp), w.shape)
    ogw, ogb = ocpu.conv2d_bwd_wb(xp, gp, w.shape)
    assert relinf(gw.cpu().numpy(), ogw) < TOL
    assert relinf(gb.cpu().numpy(), ogb) < TOL
    # and on the cancelling +-1 inputs: every entry within 1e-5 of the sum of MAGNITUDES it is made
    # of (the honest bound for a sum that cancels; the float64 oracle supplies it)
    gw2, gb2 = ops.conv2d_backward_wb(_dev(x), _dev(gy), w.shape)
    ogw2, ogb2 = ocpu.conv2d_bwd_wb(x, gy, w.shape)
    mag_w, mag_b = ocpu.conv2d_bwd_wb(np.abs(x), np.abs(gy), w.shape)
    assert (np.abs(gw2.cpu().numpy() - ogw2) <= TOL * mag_w).all()
    assert (np.abs(gb2.cpu().numpy() - ogb2) <= TOL * mag_b).all()


@pytest.mark.parametrize("shape", [
    (4, 32, 140, 330, 32, 3, 3),   # the LDS-ring kernel (conv_ring.hip): BASELINE configs[3]'s channel counts
    (5, 20, 100, 300, 14, 3, 3),   # ring, ragged on both sides (input gradient: 14 -> 20 channels)
    (3, 16, 150, 200, 35, 3, 3),   # row kernels (three channel tiles on the forward side)
])
def test_conv_fused_relu_paths_vs_oracle(shape):
    """Forward with the ReLU fused into the store (newConvolution.py:150-159) and the input gradient
    with the ReLU backward of the layer below fused in (loss.backward(), :184), on shapes large
    enough for the ring / row kernels."""
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    N, Cin, H, W, Cout, kH, kW = shape
    x = synth.uniform(90, (N, Cin, H, W), -1, 1)
    w = synth.uniform(91, (Cout, Cin, kH, kW), -1, 1)
    b = synth.uniform(92, (Cout,), -1, 1)
    gy = synth.uniform(93, (N, Cout, H - kH + 1, W - kW + 1), -1, 1)
    below = synth.uniform(94, (N, Cin, H, W), -1, 1)          # the layer below's post-ReLU output
    below = np.maximum(below, 0.0).astype(np.float32)
    y = ops.conv2d_forward(_dev(x), _dev(w), _dev(b), relu=True).cpu().numpy()
    ref = ocpu.conv2d_fwd(x, w, b)
    assert relinf(y, np.maximum(ref, 0.0)) < TOL
    # entries whose pre-activation is within rounding of zero may clip either way; none may differ
    # from the oracle's clip by more than the pre-activation's own rounding error
    gx = ops.conv2d_backward_data(_dev(gy), _dev(w), x.shape, relu_mask=_dev(below)).cpu().numpy()
    refg = ocpu.conv2d_bwd_data(gy, w, x.shape) * (below > 0)
    assert relinf(gx, refg) < TOL
    assert (gx[below <= 0] == 0).all()


def test_conv_ring_forward_and_masked_paths_in_a_subprocess():
    """conv_ring.hip by default takes only the unmasked input gradient (where it is the faster
    kernel); its forward and fused-mask variants are reached with WM_CONV_RING=all, which the
    library reads once per process -- so they are checked in ONE child process
    (tests/ring_paths_check.py: the ring shapes of the tests above against the float64 oracle)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, WM_CONV_RING="all")
    r = subprocess.run([sys.executable, os.path.join(here, "ring_paths_check.py")], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ring paths ok" in r.stdout


def test_conv_autograd_matches_torch_semantics():
    """loss.backward() through Conv2dFunction (the reference call site newConvolution.py:184)."""
    from weather_model_amd import ops, synth
    from oracle import cpu as ocpu
    x = synth.uniform(60, (2, 1, 21, 40))
    w = synth.uniform(61, (1, 1, 3, 3), -0.3, 0.3)
    b = synth.uniform(62, (1,), -0.3, 0.3)
    xt = _dev(x).requires_grad_(True)
    wt = _dev(w).requires_grad_(True)
    bt = _dev(b).requires_grad_(True)
    y = ops.conv2d(xt, wt, bt)
    loss = (y * y).mean()
    loss.backward()
    yo = ocpu.conv2d_fwd(x, w, b)
    gy = (2.0 * yo / yo.size).astype(np.float32)
    ogw, ogb = ocpu.conv2d_bwd_wb(x, gy, w.shape)
    assert relinf(wt.grad.cpu().numpy(), ogw) < 2e-5
    assert relinf(bt.grad.cpu().numpy(), ogb) < 2e-5
    assert relinf(xt.grad.cpu().numpy(), ocpu.conv2d_bwd_data(gy, w, x.shape)) < 2e-5
    # unbatched (C,H,W) input, as the reference passes it (newConvolution.py:149)
    y3 = ops.conv2d(_dev(x)[0], _dev(w), _dev(b))
    assert y3.shape == (1, 19, 38)
    assert torch.equal(y3, y[0].detach())


def test_conv_bwd_wb_is_bitwise_reproducible():
    from weather_model_amd import ops, synth
    x = _dev(synth.uniform(70, (4, 1, 200, 300)))
    gy = _dev(synth.uniform(71, (4, 1, 198, 298), -1, 1))
    a = ops.conv2d_backward_wb(x, gy, (1, 1, 3, 3))
    b = ops.conv2d_backward_wb(x, gy, (1, 1, 3, 3))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("shape", [
    (4, 1, 200, 300, 1, 3, 3),    # flat stencil kernel: entry-major partials
    (64, 1, 721, 1440, 1, 3, 3),  # BASELINE configs[2] at full size
    (5, 1, 40, 66, 1, 3, 3),      # strip stencil kernel (W % 4 != 0): part-major partials
    (1, 1, 4, 4100, 1, 3, 3),     # too wide for the flat kernels
    (2, 3, 8, 9, 2, 2, 3),        # generic kernel
    (3, 16, 12, 20, 1, 3, 1),     # the reference's conv2 orientation transposed: Cout = 1
])
def test_conv_bwd_wb_deferred_finalize_is_bit_identical(shape):
    """wm_conv2d_bwd_wb_partials_f32 + wm_partials_finish_f32 == wm_conv2d_bwd_wb_f32, and
    + wm_adam_step_partials_f32 == wm_conv2d_bwd_wb_f32 + wm_adam_step_f32, bit for bit, over
    three chained steps (newConvolution.py:184-186)."""
    from weather_model_amd import ops, synth
    N, Cin, H, W, Cout, kH, kW = shape
    wshape = (Cout, Cin, kH, kW)
    x = _dev(synth.uniform(80, (N, Cin, H, W)))
    gy = _dev(synth.uniform(81, (N, Cout, H - kH + 1, W - kW + 1), -1, 1))
    gw, gb = ops.conv2d_backward_wb(x, gy, wshape)
    part = ops.conv2d_backward_wb_partials(x, gy, wshape)
    gw2, gb2 = part.finish()
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gw3, none = part.finish(need_bias=False)
    assert none is None and torch.equal(gw, gw3)
    nent = gw.numel() + gb.numel()
    p0 = _dev(synth.uniform(82, (nent,), -0.5, 0.5))
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    gout = torch.empty_like(p0)
    for step in (1, 2, 3):
        xs = x * float(step)
        g = torch.cat([t.reshape(-1) for t in ops.conv2d_backward_wb(xs, gy, wshape)])
        ops.adam_step(pa, g, ma, va, step, lr=1e-2, grad_scale=0.5)
        part = ops.conv2d_backward_wb_partials(xs, gy, wshape, ws=part.ws)
        ops.adam_step_partials(pb, part, mb, vb, step, lr=1e-2, grad_scale=0.5, grads_out=gout)
        assert torch.equal(g, gout)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert not torch.equal(pa, p0)


def test_conv_bwd_wb_deferred_rejects_mfma_shapes():
    from weather_model_amd import ops, _lib
    x = torch.zeros(2, 16, 200, 400, device="cuda")
    gy = torch.zeros(2, 24, 198, 398, device="cuda")
    with pytest.raises(_lib.WmError, match="not deferrable"):
        ops.conv2d_backward_wb_partials(x, gy, (24, 16, 3, 3))


def test_conv_error_behaviour():
    from weather_model_amd import ops, _lib
    x = torch.zeros(1, 2, 5, 5, device="cuda")
    with pytest.raises(RuntimeError, match="channels"):
        ops.conv2d(x, torch.zeros(1, 3, 3, 3, device="cuda"))
    with pytest.raises(RuntimeError, match="greater than input"):
        ops.conv2d(x, torch.zeros(1, 2, 7, 7, device="cuda"))
    with pytest.raises(NotImplementedError):
        ops.conv2d(x, torch.zeros(1, 2, 3, 3, device="cuda"), padding=1)
    L = _lib.lib()
    rc = L.wm_conv2d_bwd_wb_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, 0, 0,
                                1, 2, 5, 5, 1, 3, 3, 0)
    assert rc == -3 and b"workspace" in L.wm_last_error()
