"""GPU: the multi-layer Conv2d + ReLU stack (BASELINE configs[3] read literally) -- forward,
ReLU-masked backward-data, weight/bias gradients -- against the float64 oracle at a size the C
oracle finishes in seconds, and through size-independent properties at the full 721 x 1440 grid."""
import numpy as np
import pytest
import torch

from conftest import relinf

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _oracle_stack(x, ws, bs, gout):
    """float64 chain on the CPU oracle: activations, then gradients layer by layer."""
    from oracle import cpu as ocpu
    h = [x.astype(np.float64)]
    for w, b in zip(ws, bs):
        h.append(np.maximum(ocpu.conv2d_fwd(h[-1], w, b), 0.0))
    g = gout.astype(np.float64) * (h[-1] > 0)
    gws, gbs = [None] * len(ws), [None] * len(ws)
    for l in range(len(ws), 0, -1):
        gws[l - 1], gbs[l - 1] = ocpu.conv2d_bwd_wb(h[l - 1], g, ws[l - 1].shape)
        if l > 1:
            g = ocpu.conv2d_bwd_data(g, ws[l - 1], h[l - 1].shape) * (h[l - 1] > 0)
    return h, gws, gbs


@pytest.mark.parametrize("shape", [
    (1, 16, 260, 520, 2),     # > 131072 positions, rows >= 64 wide: the row kernels (fused mask)
    (2, 8, 30, 70, 3),        # small: split-K kernels, mask applied by the element-wise launch
])
def test_stack_matches_float64_oracle(shape):
    from weather_model_amd import synth
    from weather_model_amd.stack import ConvStack
    N, C, H, W, L = shape
    bound = 1.0 / np.sqrt(C * 9)
    x = synth.uniform(200, (N, C, H, W))
    ws = [synth.uniform(210 + l, (C, C, 3, 3), -bound, bound) for l in range(L)]
    # biases around zero so that roughly half of every layer's outputs are clipped by the ReLU
    bs = [synth.uniform(220 + l, (C,), -0.2, 0.1) for l in range(L)]
    gout = synth.uniform(230, (N, C, H - 2 * L, W - 2 * L), -1.0, 1.0)
    h, gws, gbs = _oracle_stack(x, ws, bs, gout)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()          # noqa: E731
    st = ConvStack([dev(w) for w in ws], [dev(b) for b in bs], x.shape)
    out = st.forward(dev(x))
    for l in range(1, L + 1):
        clipped = float((h[l] == 0).mean())
        assert 0.05 < clipped < 0.95, (l, clipped)        # the mask is really exercised
        assert relinf(st.h[l].cpu().numpy(), h[l]) < TOL
    assert out is st.h[L]
    flat = st.backward(dev(gout))
    for l in range(L):
        assert relinf(st.gw[l].cpu().numpy(), gws[l]) < TOL, l
        assert relinf(st.gb[l].cpu().numpy(), gbs[l]) < TOL, l
    # the flat buffer is [dW_1..dW_L, db_1..db_L] and the views alias it
    ref_flat = np.concatenate([g.reshape(-1) for g in gws] + [g.reshape(-1) for g in gbs])
    assert relinf(flat.cpu().numpy(), ref_flat) < TOL


def test_fused_relu_mask_equals_two_launches_full_size():
    """(8,32,721,1440) x Conv(32->32,3x3): backward-data with the mask fused into its store equals
    backward-data followed by the element-wise mask; masked positions are exact zeros;
    the weight gradient is linear in the upstream gradient."""
    from weather_model_amd import ops
    N, C, H, W = 4, 32, 721, 1440
    w = ops.synth_uniform((C, C, 3, 3), seed=301, lo=-1.0 / 17, hi=1.0 / 17)
    g = ops.synth_uniform((N, C, H - 2, W - 2), seed=302, lo=-1.0, hi=1.0)
    ymask = ops.synth_uniform((N, C, H, W), seed=303, lo=-1.0, hi=1.0).clamp_min_(0.0)
    fused = ops.conv2d_backward_data(g, w, (N, C, H, W), relu_mask=ymask)
    plain = ops.conv2d_backward_data(g, w, (N, C, H, W))
    ops.relu_mask_(plain, ymask)
    # the two calls may take different kernels (by default the unmasked gradient of this shape goes to
    # conv_ring.hip and the masked one to the row kernel -- each where it is faster), so they agree to
    # rounding, not bit for bit; with both on one kernel the fused store IS bit for bit the two
    # launches (tests/ring_paths_check.py under WM_CONV_RING=all)
    assert float((fused - plain).abs().max()) <= 1e-5 * float(plain.abs().max())
    assert torch.equal(fused == 0, plain == 0) or float(((fused == 0) != (plain == 0)).float().mean()) < 1e-6
    assert float(fused[ymask == 0].abs().max()) == 0.0
    assert float(fused[ymask > 0].abs().max()) > 0.0


def test_stack_step_is_reproducible_and_reuses_buffers():
    from weather_model_amd import ops
    from weather_model_amd.stack import ConvStack
    N, C, H, W, L = 2, 32, 200, 300, 3
    x = ops.synth_uniform((N, C, H, W), seed=311)
    ws = [ops.synth_uniform((C, C, 3, 3), seed=312 + l, lo=-1.0 / 17, hi=1.0 / 17) for l in range(L)]
    bs = [ops.synth_uniform((C,), seed=320 + l, lo=-0.1, hi=0.1) for l in range(L)]
    g = ops.synth_uniform((N, C, H - 6, W - 6), seed=330, lo=-1.0, hi=1.0)
    st = ConvStack(ws, bs, x.shape)
    a = st.step_grads(x, g).clone()
    ptrs = [t.data_ptr() for t in st.h[1:]] + [st.flat_grads.data_ptr()]
    b = st.step_grads(x, g)
    assert torch.equal(a, b)
    assert ptrs == [t.data_ptr() for t in st.h[1:]] + [st.flat_grads.data_ptr()]
    assert st.flat_grads.numel() == L * (C * C * 9 + C)
    # doubling the upstream gradient doubles every gradient (ReLU masks do not depend on it)
    c = st.step_grads(x, 2.0 * g)
    assert float((c - 2.0 * a).abs().max()) <= 4e-6 * float(a.abs().max())
