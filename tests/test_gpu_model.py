"""GPU parity: fused reference-shaped model (through the C ABI) vs the reference's own outputs
(tests/golden/model_golden.npz, produced by running newConvolution.CustomModel) and vs the numpy
float64 oracle.  Tolerance 1e-5 relative (BASELINE.json)."""
import numpy as np
import pytest
import torch

from conftest import relinf

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _sd(g, prefix="sd0/"):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def _flat(sd):
    from weather_model_amd import ops
    return np.concatenate([np.asarray(sd[n], np.float32).reshape(-1) for n, _ in ops.MODEL_PARAM_SHAPES])


def _split(flat):
    from weather_model_amd import ops
    out, off = {}, 0
    for n, shp in ops.MODEL_PARAM_SHAPES:
        k = int(np.prod(shp))
        out[n] = flat[off:off + k].reshape(shp)
        off += k
    return out


def test_forward_matches_reference_outputs(model_golden):
    from weather_model_amd import ops
    from oracle import model_ref
    g = model_golden
    p = torch.from_numpy(_flat(_sd(g))).cuda()
    x = torch.from_numpy(g["x"][:, 0]).cuda()                      # (64,2,6)
    out = ops.model_forward(p, x).cpu().numpy()
    assert relinf(out, g["out"]) < TOL                              # the reference's own outputs
    assert relinf(out, model_ref.forward(_sd(g), g["x"][:, 0])) < TOL
    # the reference's single-sample call shape (1,2,6) gives the same row
    one = ops.model_forward(p, x[3:4]).cpu().numpy()
    assert np.array_equal(one[0], out[3])
    # ragged batch sizes (tile = 16 samples)
    for B in (1, 15, 17, 33):
        o = ops.model_forward(p, x[:B]).cpu().numpy()
        assert np.array_equal(o, out[:B])


def test_gradients_match_reference(model_golden):
    from weather_model_amd import ops
    from oracle import model_ref
    g = model_golden
    sd = _sd(g)
    p = torch.from_numpy(_flat(sd)).cuda()
    x = torch.from_numpy(g["x"][:, 0]).cuda()
    y = torch.from_numpy(g["y"][:, 0]).cuda()
    # per-sample gradients the reference produced for samples 0..3
    for i in range(4):
        out, loss, grads = ops.model_train(p, x[i:i + 1], y=y[i:i + 1])
        gs = _split(grads.cpu().numpy())
        assert abs(float(loss) - g["loss"][i]) < 1e-5 * max(1.0, g["loss"][i])
        for n in gs:
            ref = g["grad4/" + n][i]
            scale = max(np.abs(ref).max(), 1e-6)
            assert np.abs(gs[n] - ref).max() / scale < 2e-5, (i, n)
    # batch of 64 = mean of the reference's per-sample gradients
    out, loss, grads = ops.model_train(p, x, y=y)
    gs = _split(grads.cpu().numpy())
    oloss, _, ograds = model_ref.loss_and_grads(sd, g["x"][:, 0], g["y"][:, 0])
    assert abs(float(loss) - oloss) < 1e-5 * oloss
    for n in gs:
        assert relinf(gs[n], g["gradmean64/" + n]) < 2e-5, n
        assert relinf(gs[n], ograds[n]) < TOL, n
    # bitwise reproducible
    _, _, grads2 = ops.model_train(p, x, y=y)
    assert torch.equal(grads, grads2)


def test_autograd_path_and_given_dout(model_golden):
    from weather_model_amd import ops
    g = model_golden
    p = torch.from_numpy(_flat(_sd(g))).cuda().requires_grad_(True)
    x = torch.from_numpy(g["x"][:20, 0]).cuda()
    y = torch.from_numpy(g["y"][:20, 0]).cuda()
    out = ops.ModelFunction.apply(p, x)
    loss = torch.nn.functional.mse_loss(out, y)           # loss_fn at newConvolution.py:181
    loss.backward()                                        # :184
    _, loss2, grads = ops.model_train(p.detach(), x, y=y)
    assert abs(float(loss.detach()) - float(loss2)) < 1e-6
    assert relinf(p.grad.cpu().numpy(), grads.cpu().numpy()) < 1e-5


def test_adam_steps_match_reference(model_golden):
    from weather_model_amd import ops
    g = model_golden
    sd = _sd(g)
    p = torch.from_numpy(_flat(sd)).cuda()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    x = torch.from_numpy(g["x"][:, 0]).cuda()
    y = torch.from_numpy(g["y"][:, 0]).cuda()
    p0 = _flat(sd).astype(np.float64)
    prev = p0
    for step in range(3):
        _, _, grads = ops.model_train(p, x[step:step + 1], y=y[step:step + 1])
        ops.adam_step(p, grads, m, v, step + 1)
        ref = _flat(_sd(g, f"adam{step + 1}/")).astype(np.float64)
        # the UPDATE of this step, not the parameter: a parameter moves ~1e-3 per step, so an
        # absolute bound on p would let a 2 % error in the update through
        _assert_update(p.cpu().numpy().astype(np.float64) - prev, ref - prev, step)
        prev = ref
        # continue from the reference's parameters so that every step is checked on its own
        p.copy_(torch.from_numpy(_flat(_sd(g, f"adam{step + 1}/"))))
    # batched step (mean gradient over 8 samples)
    p = torch.from_numpy(_flat(sd)).cuda()
    m.zero_(); v.zero_()
    _, _, grads = ops.model_train(p, x[:8], y=y[:8])
    ops.adam_step(p, grads, m, v, 1)
    du = p.cpu().numpy().astype(np.float64) - p0
    dref = _flat(_sd(g, "adam_b8/")).astype(np.float64) - p0
    # (a) the Adam kernel itself: float64 Adam applied to OUR gradient reproduces the update
    gd = grads.cpu().numpy().astype(np.float64)
    lr, eps = 1e-3, 1e-8
    exact = -lr * gd / (np.abs(gd) + eps)          # step 1: m_hat = g, sqrt(v_hat) = |g|
    assert np.abs(du - exact).max() <= 1e-4 * lr
    # (b) against the reference's batched step.  At step 1 the update is lr * g / (|g| + eps): for
    # the few dozen weights whose mean gradient is within ~100 eps of zero it amplifies a gradient
    # difference dg by eps / (|g| + eps)^2.  The gradient itself is held to 1e-5 * ||g||_inf (the
    # conv tolerance), so an element's update is held to 1e-4 * lr plus what that dg can do to it;
    # for every weight with |g| >= 1e-5 that is 1.1e-4 * lr.
    dg = 1e-5 * np.abs(gd).max()
    bound = 1e-4 * lr + lr * eps * dg / (np.abs(gd) + eps) ** 2
    assert (np.abs(du - dref) <= bound).all(), float((np.abs(du - dref) / bound).max())
    strong = np.abs(gd) >= 1e-5
    assert strong.sum() > 10000
    assert np.abs(du - dref)[strong].max() <= 1.1e-4 * lr


def _assert_update(du, dref, tag):
    """Relative error of an Adam update, ||du - dref||_inf / ||dref||_inf <= 1e-4.  The stored
    reference parameters are float32, so a difference of two of them carries up to one ulp of the
    PARAMETER (6e-8 * |p| ~ 2e-8 for |p| ~ 0.3), i.e. ~2e-5 of a 1e-3 update: 1e-4 leaves room for
    that and nothing else."""
    scale = np.abs(dref).max()
    assert scale > 1e-4, (tag, scale)          # Adam moves parameters by ~lr = 1e-3 per step
    err = np.abs(du - dref).max() / scale
    assert err <= 1e-4, (tag, err, int(np.abs(du - dref).argmax()))


def test_large_batch_properties():
    """1e5 samples: per-sample outputs independent of batching; gradient of a doubled batch
    equals the gradient of the batch (mean semantics)."""
    from weather_model_amd import ops, synth
    from oracle import model_ref
    rng = {n: synth.uniform(500 + k, shp, -0.3, 0.3) for k, (n, shp) in enumerate(ops.MODEL_PARAM_SHAPES)}
    p = torch.from_numpy(_flat(rng)).cuda()
    B = 100_003
    x = ops.synth_uniform((B, 2, 6), seed=77)
    y = ops.synth_uniform((B, 6), seed=78)
    out = ops.model_forward(p, x)
    idx = [0, 1, 4095, 65536, B - 1]
    ref = model_ref.forward(rng, x[idx].cpu().numpy())
    assert relinf(out[idx].cpu().numpy(), ref) < TOL
    _, l1, g1 = ops.model_train(p, x, y=y)
    _, l2, g2 = ops.model_train(p, torch.cat([x, x]), y=torch.cat([y, y]))
    assert abs(float(l1) - float(l2)) < 1e-5 * float(l1)
    assert relinf(g2.cpu().numpy(), g1.cpu().numpy()) < 1e-5
    ol, _, og = model_ref.loss_and_grads(rng, x[:4096].cpu().numpy(), y[:4096].cpu().numpy())
    _, l3, g3 = ops.model_train(p, x[:4096], y=y[:4096])
    assert abs(float(l3) - ol) < 1e-5 * ol
    assert relinf(g3.cpu().numpy(), _flat(og)) < TOL


def test_dropin_module_reproduces_reference_training(model_golden):
    """weather_model_amd.dropin.newConvolution: the reference's call sequence
    (model(x); loss_fn; loss.backward(); optimizer.step(); zero_grad) on the reference's weights."""
    from weather_model_amd.dropin import newConvolution as nc
    g = model_golden
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd0/")}
    model = nc.CustomModel((2, 6))
    model.load_state_dict(sd, strict=True)
    x = torch.from_numpy(g["x"])            # (64,1,2,6) CPU tensors, as the DataLoader yields them
    y = torch.from_numpy(g["y"])
    out = model(x[0])                       # the reference call: (1,2,6) -> (1,6)
    assert out.shape == (1, 6)
    assert relinf(out.detach().cpu().numpy(), g["out"][0:1]) < TOL
    outb = model(x[:, 0])                   # batched
    assert relinf(outb.detach().cpu().numpy(), g["out"]) < TOL
    loss_fn = torch.nn.MSELoss()
    opt = torch.optim.Adam(model.parameters())
    # Three steps of the reference's loop, each held on its own UPDATE to 1e-4 (the bar of
    # test_adam_steps_match_reference), through the module path: after a step the parameters are
    # set to the reference's, the optimizer keeps its own moments.  A tensor the model never
    # touches (seq_module) must not move at all.
    prev = {k: v.numpy().astype(np.float64) for k, v in sd.items()}
    for step in range(3):
        nc.training_loop([(x[step], y[step])], model, loss_fn, opt, 1, step)
        ref = {k[len(f"adam{step + 1}/"):]: g[k] for k in g.files if k.startswith(f"adam{step + 1}/")}
        ours = {k: v.cpu().numpy().astype(np.float64) for k, v in model.state_dict().items()}
        moving = [k for k in ours if np.abs(ref[k].astype(np.float64) - prev[k]).max() > 0.0]
        for k in ours:
            if k not in moving:
                assert np.abs(ours[k] - prev[k]).max() == 0.0, k
        du = np.concatenate([(ours[k] - prev[k]).ravel() for k in moving])
        dref = np.concatenate([(ref[k].astype(np.float64) - prev[k]).ravel() for k in moving])
        _assert_update(du, dref, ("module", step))
        prev = {k: ref[k].astype(np.float64) for k in ours}
        model.load_state_dict({k: torch.from_numpy(ref[k]) for k in ours}, strict=True)
    correct, total, _ = nc.testing_loop([(x[i], y[i]) for i in range(8)], model, loss_fn)
    assert total == 8 and 0 <= correct <= 8


def test_device_trainer_matches_module_training():
    from weather_model_amd import ops
    """DeviceTrainer (gather + fused train + Adam launches) == the drop-in module trained through
    torch autograd and torch.optim.Adam on the same batches, and the loss goes down."""
    import numpy as np
    from weather_model_amd import synth
    from weather_model_amd.dropin import newConvolution as nc
    T, Z, Y, X = 9, 2, 8, 16
    data = {v: torch.from_numpy(synth.uniform(100 + k, (T, 1, Z, Y, X)))
            for k, v in enumerate(nc.data_vars)}
    shape = (T - 6, Z, Y, X)
    ds = nc.CustomDataset(data, int(np.prod(shape)), shape, 0)
    torch.manual_seed(3)
    model_a = nc.CustomModel((2, 6))
    model_b = nc.CustomModel((2, 6))
    model_b.load_state_dict(model_a.state_dict())
    trainer = nc.DeviceTrainer(model_a)
    l0 = trainer.epoch(ds, batch_size=64, shuffle=False)
    # the same batches through the module / autograd / torch Adam
    opt = torch.optim.Adam(model_b.parameters())
    loss_fn = torch.nn.MSELoss()
    losses = []
    for start in range(0, len(ds), 64):
        x, y = ds.batch(list(range(start, min(start + 64, len(ds)))), device="cuda")
        loss = loss_fn(model_b(x), y)
        losses.append(float(loss.detach()))
        loss.backward()
        opt.step()
        opt.zero_grad()
    assert l0 == pytest.approx(float(np.mean(losses)), rel=1e-4)
    for name, _ in ops.MODEL_PARAM_SHAPES:
        a = model_a.get_parameter(name).detach().cpu().numpy()
        b = model_b.get_parameter(name).detach().cpu().numpy()
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-3), name
    l1 = trainer.epoch(ds, batch_size=64, shuffle=True, seed=1)
    l2 = trainer.epoch(ds, batch_size=64, shuffle=True, seed=2)
    assert l2 < l0


def test_model_train_gather_equals_gather_then_train():
    """wm_model_train_gather_f32 (samples read from the resident volume by index) is bit for bit
    wm_gather_triplets_f32 followed by wm_model_train_f32: outputs, loss and all 20566 gradients,
    for an index list (with repeats and wrap-around) and for a contiguous range."""
    from weather_model_amd import ops, synth
    rng = {n: synth.uniform(700 + k, shp, -0.3, 0.3) for k, (n, shp) in enumerate(ops.MODEL_PARAM_SHAPES)}
    p = torch.from_numpy(_flat(rng)).cuda()
    T, Z, Y, X = 11, 3, 17, 29
    vol = ops.synth_uniform((T, Z, Y, X, 6), seed=710)
    shape = (T - 7, Z - 1, Y - 2, X - 5)
    n = int(np.prod(shape))
    gen = torch.Generator().manual_seed(3)
    idx = torch.randint(-5, 2 * n, (1003,), generator=gen).cuda()
    for kw in ({"indices": idx}, {"first": 37, "count": 2 * n + 11}):
        x, y = ops.gather_triplets(vol, shape, time_offset=1, **kw)
        o0, l0, g0 = ops.model_train(p, x, y=y)
        o1, l1, g1 = ops.model_train_gather(p, vol, shape, time_offset=1, **kw)
        assert torch.equal(o0, o1) and torch.equal(g0, g1) and float(l0) == float(l1)
    with pytest.raises(Exception, match="leaves"):
        ops.model_train_gather(p, vol, (T - 5, Z, Y, X), count=8)
