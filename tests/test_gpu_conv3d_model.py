"""GPU tests of the assembled Conv3D network (convolution.create_model) against a plain torch-CPU
restatement with the same weights evaluated in FLOAT64: forward, loss, every parameter gradient
(1e-5 in ||a-b||inf / ||b||inf, the conv tolerance of BASELINE.json), training behaviour and the
Keras-surface details.  Parity unpinned: keras/tensorflow, where the reference's network lives, are
absent, so the arbiter is the operator definition as torch-CPU float64 evaluates it."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import relinf  # noqa: E402
from weather_model_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _cpu_forward(P, x1, x2, channels_last=True):
    def pool(t):
        if channels_last:       # window over axes 1..3, axis 4 untouched
            return F.max_pool3d(t.permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1).contiguous()
        return F.max_pool3d(t, 2, 2)

    def branch(x, b):
        for li in (1, 2):
            x = pool(F.conv3d(x, P[f"{b}Conv3D{li}/kernel"], P[f"{b}Conv3D{li}/bias"]))
        return x.reshape(x.shape[0], -1)
    h = torch.cat((branch(x1, "1"), branch(x2, "2")), 1)
    h = F.relu(F.linear(h, P["dense1/kernel"], P["dense1/bias"]))
    return F.linear(h, P["output/kernel"], P["output/bias"])


def _model():
    from weather_model_amd.dropin import convolution as cv
    cv = importlib.reload(cv)
    return cv, cv.compile_model(cv.create_model((1, 20, 20, 20)))


def test_model_shapes_follow_survey_2b():
    cv, m = _model()
    shapes = {k: tuple(v.shape) for k, v in m.params.items()}
    assert shapes["1Conv3D1/kernel"] == (32, 1, 7, 7, 7)
    assert shapes["2Conv3D2/kernel"] == (64, 16, 5, 5, 5)      # 16 = 32 // 2 after the quirky pool
    assert shapes["dense1/kernel"] == (640, 640) and shapes["output/kernel"] == (8000, 640)
    assert m.metrics_names == ["loss", "mae", "msle"]
    w = m.get_weights()
    assert w[0].shape == (7, 7, 7, 1, 32) and w[-2].shape == (640, 8000)
    x = synth.uniform(1, (3, 1, 20, 20, 20))
    before = m.predict([x, x])
    assert before.shape == (3, 8000) and before.dtype == np.float32
    m2 = cv.create_model((1, 20, 20, 20))
    m2.set_weights(w)
    np.testing.assert_array_equal(m2.predict([x, x]), before)


@pytest.mark.parametrize("channels_last", [True, False])
def test_forward_and_gradients_match_torch_cpu(channels_last):
    from weather_model_amd.dropin._conv3d_model import Conv3DModel
    m = Conv3DModel((1, 20, 20, 20), seed=3, pool_channels_last=channels_last)
    m.compile()
    with torch.no_grad():                      # non-zero biases so their gradients are exercised
        for k, v in m.params.items():
            if k.endswith("bias"):
                v.copy_(torch.from_numpy(synth.uniform(hash(k) % 1000, tuple(v.shape), -0.1, 0.1)))
    B = 4
    x1 = synth.uniform(21, (B, 1, 20, 20, 20))
    x2 = synth.uniform(22, (B, 1, 20, 20, 20))
    y = synth.uniform(23, (B, 1, 8000))
    # the arbiter runs in float64 on the same float32 values
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.params.items()}
    ref = _cpu_forward(P, torch.from_numpy(x1).double(), torch.from_numpy(x2).double(), channels_last)
    got = m([x1, x2], training=True)
    assert relinf(got.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    yt = torch.from_numpy(y)
    ref_loss = ((yt.double() - ref) ** 2).mean()        # (B,1,P) - (B,P) broadcasts to (B,B,P)
    assert ((yt.double() - ref) ** 2).shape == (B, B, 8000)
    loss = m.loss_fn(got, yt.cuda())
    assert float(loss) == pytest.approx(float(ref_loss), rel=1e-5)
    ref_loss.backward()
    loss.backward()
    for k, v in m.params.items():
        assert relinf(v.grad.cpu().numpy(), P[k].grad.numpy()) < 1e-5, k


def test_training_reduces_loss_and_early_stopping():
    cv, m = _model()
    m.broadcast_targets = False
    rng = np.random.default_rng(0)
    base = rng.uniform(0.3, 0.7, size=(1, 20, 20, 20)).astype(np.float32)

    def gen():
        while True:
            a = base[None] + 0.05 * rng.standard_normal((8, 1, 20, 20, 20)).astype(np.float32)
            b = base[None] + 0.05 * rng.standard_normal((8, 1, 20, 20, 20)).astype(np.float32)
            yield a, b, (0.5 * (a + b)).reshape(8, 1, 8000)
    h = m.fit(gen, epochs=4, steps_per_epoch=6, batch_size=8,
              callbacks=[cv.EarlyStopping(monitor="loss", patience=2)], verbose=0)
    assert h.epoch == list(range(len(h.history["loss"]))) and len(h.epoch) >= 2
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert set(h.history) == {"loss", "mae", "msle"}

    class Never:                                # a monitor that never improves stops after patience
        monitor, patience = "loss", 1
    h2 = m.fit(gen, epochs=6, steps_per_epoch=1, callbacks=[Never()], verbose=0)
    assert 1 <= len(h2.epoch) <= 6
    logs = m.test_step(next(gen()))
    assert set(logs) == {"loss", "mae", "msle"}


def test_graphed_train_step_matches_eager():
    """A HIP-graph replay of the training step gives the same parameters as the eager step."""
    from weather_model_amd.dropin._conv3d_model import Conv3DModel
    a = Conv3DModel((1, 20, 20, 20), seed=5)
    b = Conv3DModel((1, 20, 20, 20), seed=5)
    metrics = ["mean_absolute_error", "mean_squared_logarithmic_error"]
    a.compile(loss="mean_squared_error", metrics=metrics)
    b.compile(loss="mean_squared_error", metrics=metrics)
    rng = np.random.default_rng(1)
    batches = [(rng.uniform(size=(4, 1, 20, 20, 20)).astype(np.float32),
                rng.uniform(size=(4, 1, 20, 20, 20)).astype(np.float32),
                rng.uniform(size=(4, 1, 8000)).astype(np.float32)) for _ in range(3)]
    la = [a.train_step(bt) for bt in batches]
    lb = [b.graphed_train_step(bt) for bt in batches]
    for key in ("loss", "mae", "msle"):                  # the metrics are computed inside the graph
        np.testing.assert_allclose([l[key] for l in la], [l[key] for l in lb], rtol=1e-5)
    for k in a.params:
        assert relinf(b.params[k].detach().cpu().numpy(), a.params[k].detach().cpu().numpy()) < 1e-5, k


def test_deferred_logs_are_the_steps_own_logs():
    """graphed_train_step(defer=True): logs read one step late (as fit() does) are bit for bit the
    logs of an immediate read, also when the next replay has already overwritten the graph's
    statistics tensor, and fit() on the deferred path gives the immediate path's history."""
    from weather_model_amd.dropin._conv3d_model import Conv3DModel
    a = Conv3DModel((1, 20, 20, 20), seed=9)
    b = Conv3DModel((1, 20, 20, 20), seed=9)
    metrics = ["mean_absolute_error", "mean_squared_logarithmic_error"]
    a.compile(loss="mean_squared_error", metrics=metrics)
    b.compile(loss="mean_squared_error", metrics=metrics)
    rng = np.random.default_rng(3)
    batches = [(rng.uniform(size=(4, 1, 20, 20, 20)).astype(np.float32),
                rng.uniform(size=(4, 1, 20, 20, 20)).astype(np.float32),
                rng.uniform(size=(4, 1, 8000)).astype(np.float32)) for _ in range(5)]
    la = [a.graphed_train_step(bt) for bt in batches]
    pend = [b.graphed_train_step(bt, defer=True) for bt in batches[:3]]     # three outstanding
    lb = [p.result() for p in pend]
    for bt in batches[3:]:
        lb.append(b.graphed_train_step(bt, defer=True).result())
    assert la == lb
    c = Conv3DModel((1, 20, 20, 20), seed=9)
    d = Conv3DModel((1, 20, 20, 20), seed=9)
    c.compile(loss="mean_squared_error", metrics=metrics)
    d.compile(loss="mean_squared_error", metrics=metrics)
    it = lambda: iter(batches)                                              # noqa: E731
    hc = c.fit(it(), epochs=1, steps_per_epoch=5, verbose=0)
    sums = {}
    for bt in batches:
        for k, v in d.graphed_train_step(bt).items():
            sums[k] = sums.get(k, 0.0) + v
    for k in sums:
        assert hc.history[k][0] == pytest.approx(sums[k] / 5, rel=1e-12)


def test_graphed_step_keeps_one_graph_per_batch_shape():
    """Alternating batch sizes (the short last batch of an epoch) replay their own graphs and stay
    on the eager trajectory."""
    from weather_model_amd.dropin._conv3d_model import Conv3DModel
    a = Conv3DModel((1, 20, 20, 20), seed=6)
    b = Conv3DModel((1, 20, 20, 20), seed=6)
    a.compile()
    b.compile()
    rng = np.random.default_rng(2)

    def batch(n):
        return (rng.uniform(size=(n, 1, 20, 20, 20)).astype(np.float32),
                rng.uniform(size=(n, 1, 20, 20, 20)).astype(np.float32),
                rng.uniform(size=(n, 1, 8000)).astype(np.float32))

    batches = [batch(n) for n in (4, 2, 4, 2)]
    la = [a.train_step(bt)["loss"] for bt in batches]
    lb = [b.graphed_train_step(bt)["loss"] for bt in batches]
    assert len(b._graphs) == 2
    np.testing.assert_allclose(la, lb, rtol=1e-5)


def test_data_generator_device_batches_bit_exact():
    """convolution.DataGenerator(data, device='cuda'): the random 20^3 cubes cut on the GPU
    (wm_gather_cubes_f32 with host-drawn origins) equal the host generator's batches bit for bit
    under the same seed (convolution.py:118-153), and the model trains from them."""
    import random
    from weather_model_amd import synth
    from weather_model_amd.dropin import convolution as cv
    data = {"T": synth.uniform(12, (9, 1, 24, 25, 27))}
    host = cv.DataGenerator(data)
    dev = cv.DataGenerator(data, device="cuda")
    random.seed(5)
    gh = host()
    hb = [next(gh) for _ in range(3)]
    random.seed(5)
    gd = dev()
    db = [next(gd) for _ in range(3)]
    for (a1, a2, ay), (b1, b2, by) in zip(hb, db):
        assert b1.is_cuda and tuple(b1.shape) == (16, 1, 20, 20, 20) and tuple(by.shape) == (16, 1, 8000)
        assert np.array_equal(a1, b1.cpu().numpy())
        assert np.array_equal(a2, b2.cpu().numpy())
        assert np.array_equal(ay, by.cpu().numpy())
    assert host.index == dev.index
    m = cv.create_model((1, 20, 20, 20))
    cv.compile_model(m)
    h = m.fit(dev, epochs=2, steps_per_epoch=2, verbose=0)
    assert len(h.history["loss"]) == 2 and np.isfinite(h.history["loss"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,P,broadcast", [(16, 8000, True), (5, 37, True), (33, 1000, True),
                                           (64, 130, True), (16, 8000, False), (1, 3, True), (200, 24, True),
                                           (4096, 6, True), (4096, 5, False), (1000, 17, True)])
def test_fused_loss_metrics_and_gradient(B, P, broadcast):
    """wm_mse_metrics_f32 against the tensor expressions of the model (keras' mean_squared_error
    on (B,1,P) targets, mae, msle with the epsilon clip) and autograd's gradient, in float64."""
    from weather_model_amd import ops
    gen = torch.Generator().manual_seed(B * 1000 + P)
    pred = (torch.rand((B, P), generator=gen) * 2.0 - 0.5)       # some values below the clip
    y = torch.rand((B, P), generator=gen) * 1.5 - 0.25
    if B >= 1000:
        # the documented batch limit: predictions near the batch mean of the targets, where the
        # gradient B*p - sum(y) cancels (float32 sums over the batch lose it: ADVICE r2)
        pred = y.mean(0, keepdim=True) + 1e-3 * (torch.rand((B, P), generator=gen) - 0.5)
    out3, grad = ops.mse_metrics(pred.cuda(), y.cuda(), broadcast)
    pd = pred.double().requires_grad_(True)
    yd = y.double()
    diff = (yd[None, :, :] - pd[:, None, :]) if broadcast else (yd - pd)
    lp = torch.log1p(pd.clamp_min(1e-7))
    ly = torch.log1p(yd.clamp_min(1e-7))
    le = (lp[:, None, :] - ly[None, :, :]) if broadcast else (lp - ly)
    ref = torch.stack([(diff ** 2).mean(), diff.abs().mean(), (le ** 2).mean()])
    (gref,) = torch.autograd.grad(ref[0], pd)
    got = out3.cpu().double()
    assert torch.allclose(got, ref.detach(), rtol=2e-6, atol=0)
    assert (grad.cpu().double() - gref).abs().max() <= 2e-6 * gref.abs().max()
    again, _ = ops.mse_metrics(pred.cuda(), y.cuda(), broadcast, want_grad=False)
    assert torch.equal(again, out3)                               # fixed-order sums
    with pytest.raises(Exception, match="out of range"):
        ops.mse_metrics(torch.zeros((4097, 4), device="cuda"), torch.zeros((4097, 4), device="cuda"), True)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(16, 640, 640), (16, 640, 8000), (4, 640, 8000), (5, 36, 20),
                                   (33, 52, 100), (1, 4, 4)])
def test_dense_layers_match_float64(M, K, N):
    """wm_dense_fwd_f32 / wm_dense_bwd_f32 (and their autograd wrapper with the ReLU backward fused
    into the consumer's gx store) against float64 torch on the CPU."""
    from weather_model_amd import ops
    gen = torch.Generator().manual_seed(M * 7 + N)
    x = torch.rand((M, K), generator=gen) - 0.5
    W1 = (torch.rand((K, K), generator=gen) - 0.5) * 0.2
    b1 = torch.rand((K,), generator=gen) - 0.5
    W2 = (torch.rand((N, K), generator=gen) - 0.5) * 0.2
    b2 = torch.rand((N,), generator=gen) - 0.5
    g = torch.rand((M, N), generator=gen) - 0.5
    leaves = [t.cuda().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    h = ops.dense(leaves[0], leaves[1], leaves[2], relu=True, relu_bwd_fused=True)
    out = ops.dense(h, leaves[3], leaves[4], relu_input=True)
    out.backward(g.cuda())
    ref = [t.double().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    hr = torch.relu(F.linear(ref[0], ref[1], ref[2]))
    outr = F.linear(hr, ref[3], ref[4])
    outr.backward(g.double())
    assert relinf(out.detach().cpu().numpy(), outr.detach().numpy()) < 1e-5
    for a, r, name in zip(leaves, ref, ("x", "W1", "b1", "W2", "b2")):
        assert relinf(a.grad.cpu().numpy(), r.grad.numpy()) < 1e-5, name
    # the contract of include/wm.h: the bias gradient rides on the weight-gradient pass
    from weather_model_amd import _lib
    L = _lib.lib()
    xs, gs, Ws = leaves[0].detach(), g.cuda(), leaves[1].detach()
    if M == 16 and N == 640:
        gbuf = torch.empty(N, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        rc = L.wm_dense_bwd_f32(xs.data_ptr(), Ws.data_ptr(), gs.data_ptr(), 0, 0, 0, gbuf.data_ptr(),
                                0, 0, M, N, K, st)
        assert rc == -1 and b"needs gW" in L.wm_last_error()
        rc = L.wm_dense_bwd_f32(xs.data_ptr(), Ws.data_ptr(), gs.data_ptr(), 0, 0, 0, 0, 0, 0, M, N, K, st)
        assert rc == -1 and b"neither" in L.wm_last_error()
    # the stand-alone form: ReLU backward applied by the layer itself
    l2 = [t.cuda().requires_grad_(True) for t in (x, W1, b1)]
    ops.dense(l2[0], l2[1], l2[2], relu=True).backward(torch.ones((M, K), device="cuda"))
    r2 = [t.double().requires_grad_(True) for t in (x, W1, b1)]
    torch.relu(F.linear(r2[0], r2[1], r2[2])).sum().backward()
    for a, r in zip(l2, r2):
        assert relinf(a.grad.cpu().numpy(), r.grad.numpy()) < 1e-5


@pytest.mark.parametrize("keras_eps", [True, False])
def test_device_adam_matches_float64_restatement_and_torch(keras_eps):
    """wm_adam_multi_f32 over several tensors (ragged sizes, one above a workgroup's 1024 elements),
    five steps, against a float64 restatement of keras.optimizers.Adam.update_step (epsilon outside
    the bias correction; convolution.py:54,185) resp. torch.optim.Adam; the torch form also against
    torch itself.  The device step counter advances by one per call."""
    from weather_model_amd import ops
    torch.manual_seed(5)
    shapes = [(7,), (33, 31), (1024,), (3, 5, 7, 11), (2049,)]
    ps = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    ref = [p.detach().double().cpu().numpy().copy() for p in ps]
    m = [np.zeros_like(r) for r in ref]
    v = [np.zeros_like(r) for r in ref]
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-7
    opt = ops.DeviceAdam(ps, lr=lr, betas=(b1, b2), eps=eps, keras_eps=keras_eps)
    tps = [p.detach().clone().requires_grad_(True) for p in ps]
    topt = torch.optim.Adam(tps, lr=lr, betas=(b1, b2), eps=eps)
    for t in range(1, 6):
        grads = [torch.randn(s, device="cuda") * (0.1 if t % 2 else 3.0) for s in shapes]
        for p, q, g in zip(ps, tps, grads):
            p.grad = g.clone()
            q.grad = g.clone()
        opt.step()
        topt.step()
        for k, g in enumerate(grads):
            gd = g.double().cpu().numpy()
            m[k] += (gd - m[k]) * (1 - b1)
            v[k] += (gd * gd - v[k]) * (1 - b2)
            if keras_eps:
                alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
                ref[k] -= m[k] * alpha / (np.sqrt(v[k]) + eps)
            else:
                ref[k] -= lr / (1 - b1 ** t) * m[k] / (np.sqrt(v[k]) / np.sqrt(1 - b2 ** t) + eps)
        assert int(opt._step.item()) == t
    for k, p in enumerate(ps):
        moved = np.abs(ref[k] - tps[k].detach().double().cpu().numpy()).max() if keras_eps else None
        err = np.abs(p.detach().double().cpu().numpy() - ref[k]).max()
        assert err < 2e-6 * max(1.0, np.abs(ref[k]).max()), (k, err)
        if not keras_eps:
            assert np.abs((p - tps[k]).detach().cpu().numpy()).max() < 2e-6, k
        else:
            assert moved < 1e-3       # the two epsilon placements differ only where sqrt(v) ~ eps


def test_device_adam_overlap_from_gradient_hooks_equals_plain_step():
    """DeviceAdam.overlap: tensors updated from their gradient hooks on a side stream while the
    backward pass is still running, the rest in step() -- bit for bit the one-launch update, step
    counter advanced once, eager and replayed from a captured graph."""
    from weather_model_amd import ops
    torch.manual_seed(11)
    shapes = [(640, 96), (96,), (96, 40), (40,)]
    init = [torch.randn(s, device="cuda") * 0.1 for s in shapes]
    x = torch.randn((16, 640), device="cuda")
    y = torch.randn((16, 40), device="cuda")

    def make(early):
        ps = [t.clone().requires_grad_(True) for t in init]
        opt = ops.DeviceAdam(ps, lr=1e-2, keras_eps=True)
        if early:
            opt.overlap([ps[2], ps[0]])
        return ps, opt

    def one_step(ps, opt, xx):
        opt.zero_grad(set_to_none=True)
        h = torch.relu(xx @ ps[0] + ps[1])
        loss = ((h @ ps[2] + ps[3] - y) ** 2).mean()
        loss.backward()
        opt.step()
        return loss.detach()

    a, oa = make(False)
    b, ob = make(True)
    for t in range(1, 4):
        la, lb = one_step(a, oa, x), one_step(b, ob, x)
        assert float(la) == float(lb)
        for p, q in zip(a, b):
            assert torch.equal(p, q)
        assert int(ob._step.item()) == t
    with pytest.raises(ValueError):
        ob.overlap([b[1], b[3]])                  # something must stay in step()
    # the same under graph capture: hooks fire during capture, the side stream joins before step()
    c, oc = make(True)
    xs = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        one_step(c, oc, xs)
    torch.cuda.current_stream().wait_stream(side)
    oc.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        one_step(c, oc, xs)
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    assert int(oc._step.item()) == 3            # one warm-up step + two replays
    for p, q in zip(a, c):
        assert torch.equal(p, q)


@pytest.mark.parametrize("shape", [(16, 64, 3, 3, 10), (3, 4, 6, 5, 7), (2, 2, 2, 2, 1), (5, 9, 8, 4, 14)])
def test_maxpool_cat_pair_equals_pools_and_concatenation(shape):
    """ops.maxpool3d_cat (one launch for both branches' pools, written straight into the concatenated
    matrix; convolution.py:164-174) is bit for bit cat(pool(a), pool(b)) -- values against
    torch.nn.functional.max_pool3d too -- and so are its input gradients."""
    from weather_model_amd import ops
    torch.manual_seed(3)
    a = torch.randn(shape, device="cuda").requires_grad_(True)
    b = torch.randn(shape, device="cuda").requires_grad_(True)
    h = ops.maxpool3d_cat(a, b)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = torch.cat((ops.maxpool3d(a2, True).flatten(1), ops.maxpool3d(b2, True).flatten(1)), 1)
    assert torch.equal(h, ref)
    tp = F.max_pool3d(a.detach().permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1).flatten(1)
    assert torch.equal(h[:, :h.shape[1] // 2], tp)
    g = torch.randn_like(h)
    h.backward(g)
    ref.backward(g)
    assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad)
    assert float(a.grad.abs().sum()) > 0


def test_copy_multi_one_launch():
    from weather_model_amd import ops
    torch.manual_seed(4)
    srcs = [torch.randn(16, 1, 20, 20, 20, device="cuda"), torch.randn(7, device="cuda"),
            torch.arange(1001, device="cuda", dtype=torch.uint8)[1:],          # misaligned bytes
            torch.randn(3, 5, device="cuda", dtype=torch.float64)]
    dsts = [torch.zeros_like(t) for t in srcs]
    ops.copy_multi(dsts, srcs)
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    many_s = [torch.randn(33 + k, device="cuda") for k in range(11)]          # more than one launch
    many_d = [torch.empty_like(t) for t in many_s]
    ops.copy_multi(many_d, many_s)
    assert all(torch.equal(d, s) for d, s in zip(many_d, many_s))
    with pytest.raises(ValueError):
        ops.copy_multi([dsts[0]], [srcs[1]])
    with pytest.raises(ValueError):
        ops.copy_multi([torch.zeros(4, 4, device="cuda").t()], [torch.zeros(4, 4, device="cuda")])
