"""CPU: pin the oracle (oracle/) against the golden fixtures produced from the reference."""
import numpy as np
import pytest

from conftest import relinf
from cases import CONV_CASES, REG_CASES, conv_inputs, grid721_inputs
from oracle import cpu as ocpu
from oracle import model_ref
from weather_model_amd import synth


@pytest.mark.parametrize("ci", range(len(CONV_CASES)), ids=[c[0] for c in CONV_CASES])
def test_conv_oracle_matches_torch_golden(ci, conv_golden):
    name = CONV_CASES[ci][0]
    x, w, b, gy = conv_inputs(ci)
    g = conv_golden
    assert relinf(ocpu.conv2d_fwd(x, w, b), g[f"{name}/y_f64"]) < 1e-13
    assert relinf(ocpu.conv2d_bwd_data(gy, w, x.shape), g[f"{name}/gx_f64"]) < 1e-13
    gw, gb = ocpu.conv2d_bwd_wb(x, gy, w.shape)
    assert relinf(gw, g[f"{name}/gw_f64"]) < 1e-12
    assert relinf(gb, g[f"{name}/gb_f64"]) < 1e-12
    # second opinion: the library's own float32 result is within float32 rounding of the oracle
    assert relinf(g[f"{name}/y_f32"], g[f"{name}/y_f64"]) < 1e-5


def test_conv_oracle_full_grid(conv_golden):
    g = conv_golden
    x, w, b, gy = grid721_inputs()
    y = ocpu.conv2d_fwd(x, w, b)
    assert abs(y.sum() - g["grid721/y_sum"]) < 1e-9 * abs(g["grid721/y_sum"])
    assert abs((y ** 2).sum() - g["grid721/y_sumsq"]) < 1e-9 * g["grid721/y_sumsq"]
    assert relinf(y[:, :, :8, :8], g["grid721/y_tl"]) < 1e-13
    assert relinf(y[:, :, -8:, -8:], g["grid721/y_br"]) < 1e-13
    assert relinf(y[:, :, :8, -8:], g["grid721/y_tr"]) < 1e-13
    assert relinf(y[:, :, 355:363, 715:723], g["grid721/y_mid"]) < 1e-13
    gw, gb = ocpu.conv2d_bwd_wb(x, gy, w.shape)
    assert relinf(gw, g["grid721/gw"]) < 1e-11
    assert relinf(gb, g["grid721/gb"]) < 1e-10
    gx = ocpu.conv2d_bwd_data(gy, w, x.shape)
    assert abs((gx ** 2).sum() - g["grid721/gx_sumsq"]) < 1e-9 * g["grid721/gx_sumsq"]
    assert relinf(gx[:, :, :8, :8], g["grid721/gx_tl"]) < 1e-13
    assert relinf(gx[:, :, -8:, -8:], g["grid721/gx_br"]) < 1e-13


def _sd(model_golden, prefix="sd0/"):
    return {k[len(prefix):]: model_golden[k] for k in model_golden.files if k.startswith(prefix)}


def test_model_oracle_forward_matches_reference(model_golden):
    g = model_golden
    sd = _sd(g)
    out = model_ref.forward(sd, g["x"][:, 0])
    assert relinf(out, g["out"]) < 2e-6          # reference ran in float32
    assert relinf(model_ref.calculate_average(g["x"][:, 0]), g["average"]) < 1e-6


def test_model_oracle_grads_match_reference(model_golden):
    g = model_golden
    sd = _sd(g)
    for i in range(4):
        loss, per, grads = model_ref.loss_and_grads(sd, g["x"][i], g["y"][i])
        assert abs(loss - g["loss"][i]) < 2e-6 * max(1.0, abs(g["loss"][i]))
        for n in model_ref.PARAM_NAMES:
            assert relinf(grads[n], g["grad4/" + n][i]) < 5e-6, n
    loss, per, grads = model_ref.loss_and_grads(sd, g["x"][:, 0], g["y"][:, 0])
    assert relinf(per, g["loss"]) < 2e-6
    for n in model_ref.PARAM_NAMES:
        assert relinf(grads[n], g["gradmean64/" + n]) < 5e-6, n


def test_model_oracle_adam_matches_reference(model_golden):
    g = model_golden
    p = {k: v.astype(np.float64) for k, v in _sd(g).items()}
    opt = model_ref.Adam(p)
    for step in range(3):
        _, _, grads = model_ref.loss_and_grads(p, g["x"][step], g["y"][step])
        new = opt.step(p, grads)
        p.update({k: new[k] for k in model_ref.PARAM_NAMES})
        ref = _sd(g, f"adam{step + 1}/")
        for n in model_ref.PARAM_NAMES:
            # Adam's first steps move every weight by ~lr regardless of gradient size, so a
            # float32-vs-float64 sign flip of a ~0 gradient is possible only for exact zeros
            # (dead ReLU paths); compare on the update.
            assert np.max(np.abs(p[n] - ref[n])) < 2e-5, (step, n)
        # seq_module is never touched by the optimiser (newConvolution.py:113-127 is dead code)
        for k in ref:
            if k.startswith("seq_module"):
                assert np.array_equal(ref[k], g["sd0/" + k])
    # batched step (mean gradient over 8 samples)
    p = {k: v.astype(np.float64) for k, v in _sd(g).items()}
    opt = model_ref.Adam(p)
    _, _, grads = model_ref.loss_and_grads(p, g["x"][:8, 0], g["y"][:8, 0])
    new = opt.step(p, grads)
    ref = _sd(g, "adam_b8/")
    for n in model_ref.PARAM_NAMES:
        assert np.max(np.abs(new[n] - ref[n])) < 2e-5, n


@pytest.mark.parametrize("case", REG_CASES, ids=[c[0] for c in REG_CASES])
def test_regression_oracle_matches_lstsq(case, regression_golden):
    name, n, F, P, seed, dist = case
    X, Y, _ = synth.regression_problem(n, F, P, seed=seed, dist=dist)
    acc = ocpu.gram(X, Y)
    D = F + 1 + P
    assert acc.shape == (D, D)
    assert acc[F, F] == n
    assert np.allclose(acc, acc.T, rtol=0, atol=0)
    tr = regression_golden[f"{name}/gram_trace"]
    assert abs(np.trace(acc) - tr) < 1e-10 * tr
    beta = ocpu.normal_solve(acc, F, P)
    ref = regression_golden[f"{name}/beta_lstsq"]
    assert relinf(beta, ref) < 1e-7
    assert relinf(ocpu.lstsq(X, Y), ref) < 1e-9


def test_synth_is_counter_based():
    a = synth.uniform(5, (100, 16))
    b = synth.uniform(5, (40, 16), offset=60 * 16)
    assert np.array_equal(a[60:], b)
    assert a.min() >= 0.0 and a.max() < 1.0
    g = synth.normal(6, (200000,))
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01


def test_eager_torch_restatement_matches_reference(model_golden):
    """oracle/model_torch.py (bench.py's library CPU baseline for the model) against the reference's
    own run: outputs, per-sample gradients and three Adam steps.  Same library, same float32
    arithmetic, same operation order -> agreement at rounding level."""
    import torch
    from oracle import model_torch
    g = model_golden
    sd = _sd(g)
    torch.manual_seed(0)
    m = model_torch.EagerModel(sd)
    x = torch.from_numpy(g["x"])
    y = torch.from_numpy(g["y"])
    with torch.no_grad():
        out = torch.cat([m(x[i]) for i in range(8)]).numpy()
    assert relinf(out, g["out"][:8]) < 1e-6
    for i in range(2):
        m.zero_grad()
        torch.nn.functional.mse_loss(m(x[i]), y[i]).backward()
        for n in model_torch.PARAM_NAMES:
            assert relinf(m.get(n).grad.numpy(), g["grad4/" + n][i]) < 2e-6, n
    m.zero_grad()
    losses = model_torch.train_steps(m, x[:3], y[:3])
    assert np.allclose(losses[0], g["loss"][0], rtol=1e-5)
    ref = _sd(g, "adam3/")
    for n in model_torch.PARAM_NAMES:
        moved = ref[n].astype(np.float64) - sd[n].astype(np.float64)
        ours = m.get(n).detach().numpy().astype(np.float64) - sd[n].astype(np.float64)
        assert np.abs(ours - moved).max() <= 1e-3 * max(np.abs(moved).max(), 1e-12), n
