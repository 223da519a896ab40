"""GPU parity of the 3-D operators through the C ABI: wm_conv3d_* and
wm_maxpool3d_* against the float64 C oracle on the same seeded inputs, and against the committed
torch-CPU vectors.  Tolerance 1e-5 relative (max-norm over the tensor), as for conv2d."""
import os

import numpy as np
import pytest
import torch

from cases import CONV3D_CASES, CONV3D_LARGE_CASES, conv3d_inputs
from conftest import relinf
from weather_model_amd import ops, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5


@pytest.fixture(scope="module")
def ocpu():
    from oracle import cpu
    cpu.build()
    return cpu


@pytest.fixture(scope="module")
def g3():
    return np.load(os.path.join(HERE, "golden", "conv3d_golden.npz"))


def _dev(*arrays):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrays]


@pytest.mark.parametrize("case", CONV3D_CASES, ids=[c[0] for c in CONV3D_CASES])
def test_conv3d_forward_backward_vs_oracle(case, ocpu, g3):
    name, stride = case[0], case[7]
    x, w, b, gy = conv3d_inputs(case)
    xd, wd, bd, gyd = _dev(x, w, b, gy)
    y = ops.conv3d_forward(xd, wd, bd).cpu().numpy()
    assert relinf(y, ocpu.conv3d_fwd(x, w, b)) < TOL
    gx = ops.conv3d_backward_data(gyd, wd, x.shape).cpu().numpy()
    assert relinf(gx, ocpu.conv3d_bwd_data(gy, w, x.shape)) < TOL
    gw, gb = ops.conv3d_backward_wb(xd, gyd, w.shape)
    ow, ob = ocpu.conv3d_bwd_wb(x, gy, w.shape)
    assert relinf(gw.cpu().numpy(), ow) < TOL
    assert relinf(gb.cpu().numpy(), ob) < TOL
    # and the committed torch-CPU float64 vectors
    for key, val in (("y", y), ("gx", gx), ("gw", gw.cpu().numpy()), ("gb", gb.cpu().numpy())):
        ref = g3[f"{name}/{key}"]
        got = val.ravel()[::stride]
        assert np.abs(got - ref).max() / float(g3[f"{name}/{key}_absmax"]) < TOL
    # bias-free and fused-ReLU variants
    y0 = ops.conv3d_forward(xd, wd, None).cpu().numpy()
    assert relinf(y0, ocpu.conv3d_fwd(x, w, None)) < TOL
    yr = ops.conv3d_forward(xd, wd, bd, relu=True).cpu().numpy()
    np.testing.assert_array_equal(yr, np.maximum(y, 0.0))


@pytest.mark.parametrize("case", CONV3D_LARGE_CASES, ids=[c[0] for c in CONV3D_LARGE_CASES])
def test_conv3d_large_grids_vs_oracle(case, ocpu):
    """The persistent row kernels (conv3d_*_rows_kernel) on grids with a depth axis."""
    x, w, b, gy = conv3d_inputs(case)
    xd, wd, bd, gyd = _dev(x, w, b, gy)
    assert relinf(ops.conv3d_forward(xd, wd, bd).cpu().numpy(), ocpu.conv3d_fwd(x, w, b)) < TOL
    assert relinf(ops.conv3d_backward_data(gyd, wd, x.shape).cpu().numpy(),
                  ocpu.conv3d_bwd_data(gy, w, x.shape)) < TOL
    gw, gb = ops.conv3d_backward_wb(xd, gyd, w.shape)
    ow, ob = ocpu.conv3d_bwd_wb(x, gy, w.shape)
    assert relinf(gw.cpu().numpy(), ow) < TOL
    assert relinf(gb.cpu().numpy(), ob) < TOL
    again = ops.conv3d_backward_wb(xd, gyd, w.shape)
    assert torch.equal(gw, again[0]) and torch.equal(gb, again[1])      # fixed-order sums


def test_conv3d_bwd_wb_is_bitwise_reproducible():
    x, w, b, gy = conv3d_inputs(CONV3D_CASES[3])
    xd, gyd = _dev(x, gy)
    a = ops.conv3d_backward_wb(xd, gyd, w.shape)
    c = ops.conv3d_backward_wb(xd, gyd, w.shape)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


def test_conv3d_autograd_matches_oracle(ocpu):
    x, w, b, gy = conv3d_inputs(CONV3D_CASES[0])
    xd, wd, bd, gyd = _dev(x, w, b, gy)
    for t in (xd, wd, bd):
        t.requires_grad_(True)
    ops.conv3d(xd, wd, bd).backward(gyd)
    assert relinf(xd.grad.cpu().numpy(), ocpu.conv3d_bwd_data(gy, w, x.shape)) < TOL
    ow, ob = ocpu.conv3d_bwd_wb(x, gy, w.shape)
    assert relinf(wd.grad.cpu().numpy(), ow) < TOL and relinf(bd.grad.cpu().numpy(), ob) < TOL


def test_conv3d_rejects_bad_shapes():
    x = torch.zeros((1, 2, 4, 4, 4), device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv3d_forward(x, torch.zeros((3, 1, 2, 2, 2), device="cuda"))       # channel mismatch
    with pytest.raises(RuntimeError):
        ops.conv3d_forward(x, torch.zeros((3, 2, 5, 2, 2), device="cuda"))       # kernel > input
    with pytest.raises(Exception):
        ops.conv3d_forward(x, torch.zeros((65, 2, 2, 2, 2), device="cuda"))      # > 64 channels
    with pytest.raises(RuntimeError):
        ops.conv3d_forward(x.cpu(), torch.zeros((3, 2, 2, 2, 2)))                # no CPU path


def test_maxpool3d_vs_oracle_and_golden(ocpu, g3):
    xp = synth.uniform(361, (2, 32, 14, 14, 14), -1.0, 1.0)
    y, which = ops.maxpool3d_forward(torch.from_numpy(xp).cuda())
    np.testing.assert_array_equal(y.cpu().numpy(), g3["pool/y"])
    oy, ow = ocpu.maxpool3d_fwd(xp)
    np.testing.assert_array_equal(which.cpu().numpy(), ow)
    xo = synth.uniform(362, (1, 5, 7, 4, 3), -1.0, 1.0)
    xod = torch.from_numpy(xo).cuda().requires_grad_(True)
    yo = ops.maxpool3d(xod)
    np.testing.assert_array_equal(yo.detach().cpu().numpy(), g3["pool_odd/y"])
    gyo = synth.uniform(363, tuple(yo.shape), -1.0, 1.0)
    yo.backward(torch.from_numpy(gyo).cuda())
    np.testing.assert_array_equal(xod.grad.cpu().numpy(), g3["pool_odd/gx"])
    # channels_first variant = torch's own layout
    xc = synth.uniform(364, (2, 3, 6, 4, 8), -1.0, 1.0)
    yc, _ = ops.maxpool3d_forward(torch.from_numpy(xc).cuda(), channels_last=False)
    ref = torch.nn.functional.max_pool3d(torch.from_numpy(xc), 2, 2).numpy()
    np.testing.assert_array_equal(yc.cpu().numpy(), ref)
