"""The 3-D oracle (oracle/wm_oracle.c conv3d_* / maxpool3d_*) against the committed torch-CPU
float64 vectors (tests/golden/conv3d_golden.npz).  Tolerance 1e-12 relative: both sides sum in
float64 and differ only in summation order."""
import os

import numpy as np
import pytest

from cases import CONV3D_CASES, conv3d_inputs
from weather_model_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g3():
    return np.load(os.path.join(HERE, "golden", "conv3d_golden.npz"))


@pytest.fixture(scope="module")
def ocpu():
    from oracle import cpu
    cpu.build()
    return cpu


def _check(g3, name, key, val, stride, tol=1e-12):
    flat = np.asarray(val, dtype=np.float64).ravel()
    ref = g3[f"{name}/{key}"]
    scale = float(g3[f"{name}/{key}_absmax"]) or 1.0
    assert np.abs(flat[::stride] - ref).max() / scale < tol
    assert abs(flat.sum() - float(g3[f"{name}/{key}_sum"])) / (scale * np.sqrt(flat.size)) < 1e-10


@pytest.mark.parametrize("case", CONV3D_CASES, ids=[c[0] for c in CONV3D_CASES])
def test_conv3d_oracle_matches_torch_cpu(case, g3, ocpu):
    name, stride = case[0], case[7]
    x, w, b, gy = conv3d_inputs(case)
    _check(g3, name, "y", ocpu.conv3d_fwd(x, w, b), stride)
    _check(g3, name, "gx", ocpu.conv3d_bwd_data(gy, w, x.shape), stride)
    gw, gb = ocpu.conv3d_bwd_wb(x, gy, w.shape)
    _check(g3, name, "gw", gw, stride)
    _check(g3, name, "gb", gb, stride)


def test_maxpool3d_oracle_matches_torch_cpu(g3, ocpu):
    xp = synth.uniform(361, (2, 32, 14, 14, 14), -1.0, 1.0)
    y, which = ocpu.maxpool3d_fwd(xp)
    np.testing.assert_array_equal(y, g3["pool/y"])
    assert which.max() <= 7
    xo = synth.uniform(362, (1, 5, 7, 4, 3), -1.0, 1.0)
    yo, wo = ocpu.maxpool3d_fwd(xo)
    np.testing.assert_array_equal(yo, g3["pool_odd/y"])
    gyo = synth.uniform(363, yo.shape, -1.0, 1.0)
    np.testing.assert_array_equal(ocpu.maxpool3d_bwd(gyo, wo, xo.shape), g3["pool_odd/gx"])
