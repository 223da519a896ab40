#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONCE, in the build container).

What it runs
  * the reference itself: ``/root/reference/newConvolution.py`` is imported (read-only, no
    bytecode written) with a stub for the one missing import, ``xarray``, which that file only
    touches inside ``load_data()`` (newConvolution.py:6,66).  Its ``CustomModel``,
    ``CustomDataset``, ``calculate_average``, ``normalize`` and ``append_to_array`` are
    executed on torch-CPU and their inputs/outputs stored as data.
  * the library operator the reference calls for its convolutions,
    ``torch.nn.functional.conv2d`` (+ autograd), in float32 and float64 on CPU.
  * ``numpy.linalg.lstsq`` (float64) for the regression problems -- NOT the reference
    (tensorflow/keras are absent here), hence "parity unpinned" for that path.

Nothing from /root/reference is copied: the fixtures hold only arrays.  The GPU box has no
/root/reference; tests read the .npz files only.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from weather_model_amd import synth  # noqa: E402

REF = "/root/reference"


def import_reference(name):
    sys.dont_write_bytecode = True
    if "xarray" not in sys.modules:
        m = types.ModuleType("xarray")
        m.__spec__ = importlib.machinery.ModuleSpec("xarray", None)
        sys.modules["xarray"] = m
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def model_goldens():
    ref = import_reference("newConvolution")
    torch.manual_seed(20241008)
    torch.set_num_threads(1)
    model = ref.CustomModel((2, 6))
    sd0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    nS = 64
    x = synth.uniform(11, (nS, 1, 2, 6))
    y = synth.uniform(12, (nS, 1, 6))
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    loss_fn = torch.nn.MSELoss()
    names = [n for n, _ in model.named_parameters() if not n.startswith("seq_module")]
    out = np.zeros((nS, 6), np.float32)
    loss = np.zeros((nS,), np.float32)
    grads = {n: [] for n in names}
    gsum = {n: 0.0 for n in names}
    for i in range(nS):
        model.zero_grad()
        pred = model(xt[i])                       # (1,2,6) in, (1,6) out -- the reference call
        l = loss_fn(pred, yt[i])
        l.backward()
        out[i] = pred.detach().numpy()[0]
        loss[i] = l.item()
        for n, p_ in model.named_parameters():
            if n in grads:
                g = p_.grad.detach().numpy().astype(np.float64)
                gsum[n] = gsum[n] + g
                if i < 4:
                    grads[n].append(g.astype(np.float32))
    fix = {"x": x, "y": y, "out": out, "loss": loss}
    for k, v in sd0.items():
        fix["sd0/" + k] = v
    for n in names:
        fix["grad4/" + n] = np.stack(grads[n])
        fix["gradmean64/" + n] = (gsum[n] / nS)
    # seq_module never receives a gradient (it is not on the forward path)
    assert all(p_.grad is None for n, p_ in model.named_parameters() if n.startswith("seq_module"))
    # three consecutive reference training steps: backward -> Adam.step -> zero_grad (:184-186)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    opt = torch.optim.Adam(model.parameters())
    model.zero_grad()
    for step in range(3):
        pred = model(xt[step])
        l = loss_fn(pred, yt[step])
        l.backward()
        opt.step()
        opt.zero_grad()
        for k, v in model.state_dict().items():
            fix[f"adam{step + 1}/" + k] = v.detach().clone().numpy()
    # one batched step the reference cannot run itself (batch>1 raises): mean gradient
    # over the first 8 samples followed by one Adam step, emulated with the reference model by
    # accumulating per-sample grads scaled 1/8.
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    opt = torch.optim.Adam(model.parameters())
    model.zero_grad()
    for i in range(8):
        (loss_fn(model(xt[i]), yt[i]) / 8.0).backward()
    opt.step()
    for k, v in model.state_dict().items():
        fix["adam_b8/" + k] = v.detach().clone().numpy()
    fix["average"] = np.stack([ref.calculate_average(xt[i]).numpy()[0] for i in range(nS)])
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **fix)

    # CustomDataset gather (newConvolution.py:75-106) on a fake (T,1,Z,Y,X) volume per variable
    T, Z, Y, X = 9, 3, 4, 5
    vol = {v: torch.from_numpy(synth.uniform(100 + k, (T, 1, Z, Y, X)))
           for k, v in enumerate(ref.data_vars)}
    shape = (T - 6, Z, Y, X)
    ds = ref.CustomDataset(vol, int(np.prod(shape)), shape, 0)
    idxs = [0, 1, 7, 19, 59, 60, 133, len(ds) - 1]
    xs, ys = zip(*[ds[i] for i in idxs])
    np.savez_compressed(os.path.join(HERE, "dataset_golden.npz"),
                        T=T, Z=Z, Y=Y, X=X, idxs=np.array(idxs),
                        x=torch.stack(xs).numpy(), y=torch.stack(ys).numpy(),
                        length=len(ds),
                        a12=ref.append_to_array(vol["T"][0], vol["T"][1]).numpy(),
                        a123=ref.append_to_array(ref.append_to_array(vol["T"][0], vol["T"][1]),
                                                 vol["T"][2]).numpy())
    # normalize(data, var) (newConvolution.py:47-49): object with .to_numpy()
    class _Arr:
        def __init__(self, a):
            self.a = a

        def to_numpy(self):
            return self.a
    raw = synth.uniform(300, (7,), -10.0, 200.0)
    np.savez_compressed(os.path.join(HERE, "normalize_golden.npz"), raw=raw,
                        **{v: ref.normalize(_Arr(raw), v) for v in ref.data_vars})


CONV_CASES = [
    # name, N, Cin, H, W, Cout, kH, kW      (kernels/channels: SURVEY 8c item 2)
    ("ref_conv1", 3, 1, 2, 6, 16, 2, 1),        # newConvolution.py:129
    ("ref_conv2", 3, 16, 3, 6, 32, 3, 1),       # :131
    ("ref_conv3", 3, 32, 3, 6, 32, 3, 1),       # :132-133
    ("seq_1x2", 2, 8, 1, 6, 16, 1, 2),          # :116
    ("k3x3_c1", 2, 1, 37, 53, 1, 3, 3),         # BASELINE config 3 family, odd sizes
    ("k3x3_c3to5", 1, 3, 19, 23, 5, 3, 3),
    ("k3x3_c32", 1, 32, 9, 11, 32, 3, 3),       # BASELINE config 4b family
    ("k5x5", 1, 2, 13, 17, 3, 5, 5),
    ("k7x7", 1, 1, 9, 11, 2, 7, 7),
    ("edge_1x1_out", 2, 2, 3, 3, 4, 3, 3),      # H == kH, W == kW
    ("k1x1", 2, 4, 5, 7, 3, 1, 1),
    ("wide_row", 1, 1, 4, 1440, 1, 3, 3),       # a 1440-wide row: vector tails
]


def conv_goldens():
    fix = {}
    for ci, (name, N, Cin, H, W, Cout, kH, kW) in enumerate(CONV_CASES):
        fan = Cin * kH * kW
        bnd = 1.0 / np.sqrt(fan)
        x = synth.uniform(1000 + ci, (N, Cin, H, W))
        w = synth.uniform(2000 + ci, (Cout, Cin, kH, kW), -bnd, bnd)
        b = synth.uniform(3000 + ci, (Cout,), -bnd, bnd)
        gy = synth.uniform(4000 + ci, (N, Cout, H - kH + 1, W - kW + 1), -1.0, 1.0)
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            xt = torch.from_numpy(x).to(dt).requires_grad_(True)
            wt = torch.from_numpy(w).to(dt).requires_grad_(True)
            bt = torch.from_numpy(b).to(dt).requires_grad_(True)
            yt = torch.nn.functional.conv2d(xt, wt, bt, stride=1, padding=0, dilation=1, groups=1)
            gx, gw, gb = torch.autograd.grad(yt, (xt, wt, bt), torch.from_numpy(gy).to(dt))
            fix[f"{name}/y_{tag}"] = yt.detach().numpy()
            fix[f"{name}/gx_{tag}"] = gx.numpy()
            fix[f"{name}/gw_{tag}"] = gw.numpy()
            fix[f"{name}/gb_{tag}"] = gb.numpy()
        fix[f"{name}/dims"] = np.array([N, Cin, H, W, Cout, kH, kW])
        fix[f"{name}/seed_index"] = np.array(ci)
    # full-size grid of BASELINE config 3 (2 images only): checksums + corner tiles, float64
    N, H, W = 2, 721, 1440
    x = synth.uniform(2, (N, 1, H, W))
    w = synth.uniform(3, (1, 1, 3, 3), -1.0 / 3, 1.0 / 3)
    b = synth.uniform(5, (1,), -1.0 / 3, 1.0 / 3)
    gy = synth.uniform(4, (N, 1, H - 2, W - 2), -1.0, 1.0)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    bt = torch.from_numpy(b).double().requires_grad_(True)
    yt = torch.nn.functional.conv2d(xt, wt, bt)
    gx, gw, gb = torch.autograd.grad(yt, (xt, wt, bt), torch.from_numpy(gy).double())
    yn = yt.detach().numpy()
    fix["grid721/y_sum"] = np.array(yn.sum())
    fix["grid721/y_sumsq"] = np.array((yn ** 2).sum())
    fix["grid721/y_tl"] = yn[:, :, :8, :8]
    fix["grid721/y_br"] = yn[:, :, -8:, -8:]
    fix["grid721/y_tr"] = yn[:, :, :8, -8:]
    fix["grid721/y_mid"] = yn[:, :, 355:363, 715:723]
    fix["grid721/gw"] = gw.numpy()
    fix["grid721/gb"] = gb.numpy()
    gxn = gx.numpy()
    fix["grid721/gx_sum"] = np.array(gxn.sum())
    fix["grid721/gx_sumsq"] = np.array((gxn ** 2).sum())
    fix["grid721/gx_tl"] = gxn[:, :, :8, :8]
    fix["grid721/gx_br"] = gxn[:, :, -8:, -8:]
    np.savez_compressed(os.path.join(HERE, "conv_golden.npz"), **fix)


REG_CASES = [
    # name, n, F, P, seed, dist
    ("single_f1", 4096, 1, 1, 0, "uniform"),       # BASELINE config 1 (1 predictor)
    ("multi_p1", 4096, 2, 1, 21, "uniform"),       # multipleRegression p=1: (x(t), x(t+6)) -> x(t+3)
    ("f16", 65536, 16, 1, 1, "uniform"),           # config 2 family
    ("f16_normal", 65536, 16, 1, 1, "normal"),
    ("f64", 32768, 64, 1, 7, "uniform"),           # config 5 family
    ("f5_p3", 5000, 5, 3, 9, "uniform"),           # ragged n, multi-output
    ("single_p216", 2048, 432, 216, 13, "normal"),  # singleRegression p=216 shape (:74-85)
]


def regression_goldens():
    fix = {}
    for name, n, F, P, seed, dist in REG_CASES:
        X, Y, beta_true = synth.regression_problem(n, F, P, seed=seed, dist=dist)
        A = np.concatenate([X.astype(np.float64), np.ones((n, 1))], axis=1)
        beta = np.linalg.lstsq(A, Y.astype(np.float64), rcond=None)[0]
        fix[f"{name}/beta_lstsq"] = beta
        fix[f"{name}/beta_true"] = beta_true
        fix[f"{name}/cfg"] = np.array([n, F, P, seed, 0 if dist == "uniform" else 1])
        Z = np.concatenate([A, Y.astype(np.float64)], axis=1)
        fix[f"{name}/gram_trace"] = np.array(np.trace(Z.T @ Z))
    np.savez_compressed(os.path.join(HERE, "regression_golden.npz"), **fix)


CONV3D_CASES = [
    # name, B, Cin, (D,H,W), Cout, (kD,kH,kW), seed, stride of the stored subsample
    ("small", 2, 3, (6, 7, 8), 5, (3, 2, 4), 31, 1),
    ("edge_one_output", 1, 2, (3, 4, 5), 17, (3, 4, 5), 32, 1),        # D=kD..: one output each
    ("ragged_channels", 3, 33, (5, 5, 6), 18, (2, 3, 2), 33, 1),       # not multiples of 16
    ("ref_layer1", 2, 1, (20, 20, 20), 32, (7, 7, 7), 34, 5),          # convolution.py:164
    ("ref_layer2", 2, 16, (7, 7, 14), 64, (5, 5, 5), 35, 3),           # convolution.py:166 (2b)
]


def conv3d_inputs(case):
    name, B, Cin, dhw, Cout, k, seed, _ = case
    x = synth.uniform(seed * 10 + 1, (B, Cin) + dhw)
    bound = 1.0 / np.sqrt(Cin * k[0] * k[1] * k[2])
    w = synth.uniform(seed * 10 + 2, (Cout, Cin) + k, -bound, bound)
    b = synth.uniform(seed * 10 + 3, (Cout,), -bound, bound)
    out = tuple(d - kk + 1 for d, kk in zip(dhw, k))
    gy = synth.uniform(seed * 10 + 4, (B, Cout) + out, -1.0, 1.0)
    return x, w, b, gy


def conv3d_goldens():
    """torch.nn.functional.conv3d (+ autograd) and max_pool3d on CPU in float64: the library
    operator class of keras Conv3D/MaxPool3D (valid, stride 1, channels_first).  NOT the
    reference's own run (tensorflow absent) -- pins the operator definition only."""
    import torch.nn.functional as F
    fix = {}
    for case in CONV3D_CASES:
        name, stride = case[0], case[7]
        x, w, b, gy = conv3d_inputs(case)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        y = F.conv3d(xt, wt, bt)
        y.backward(torch.tensor(gy, dtype=torch.float64))
        for key, val in (("y", y.detach()), ("gx", xt.grad), ("gw", wt.grad), ("gb", bt.grad)):
            flat = val.numpy().ravel()
            fix[f"{name}/{key}"] = flat[::stride].copy()
            fix[f"{name}/{key}_sum"] = np.array(flat.sum())
            fix[f"{name}/{key}_absmax"] = np.array(np.abs(flat).max())
    # pool: channels_last window over axes 1..3, as the reference's MaxPool3D runs (2b)
    xp = synth.uniform(361, (2, 32, 14, 14, 14), -1.0, 1.0)
    yp = F.max_pool3d(torch.tensor(xp, dtype=torch.float64, requires_grad=True).permute(0, 4, 1, 2, 3),
                      2, 2).permute(0, 2, 3, 4, 1).contiguous()
    fix["pool/y"] = yp.detach().numpy().astype(np.float32)
    xo = synth.uniform(362, (1, 5, 7, 4, 3), -1.0, 1.0)                 # odd sides: floor
    xot = torch.tensor(xo, dtype=torch.float64, requires_grad=True)
    yo = F.max_pool3d(xot.permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1).contiguous()
    gyo = synth.uniform(363, tuple(yo.shape), -1.0, 1.0)
    yo.backward(torch.tensor(gyo, dtype=torch.float64))
    fix["pool_odd/y"] = yo.detach().numpy().astype(np.float32)
    fix["pool_odd/gx"] = xot.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "conv3d_golden.npz"), **fix)


def data_goldens():
    """Outputs of the reference's data.py (imported with stubs for xarray and netCDF4, which it
    only needs for file I/O) on a fake 5-file data set built by tests/fakes.py -- pins the pooled
    mean/std arithmetic and the file-name clock, quirks included."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fakes
    for name in ("xarray", "netCDF4"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    ref = import_reference("data")
    ref.X_TRAIN_DATA_PATH = "/fake/wrfout"
    n_files = 5
    scale = {v: (-(k + 1.0), 2.0 * (k + 1)) for k, v in enumerate(ref.DATA_VARS)}
    files = {ref.get_file_path(i, n_files): fakes.wrf_like_file(100 + i, ref.DATA_VARS,
                                                                scale=scale)
             for i in range(n_files)}
    sys.modules["xarray"].open_dataset = fakes.FakeXarray(files).open_dataset
    lo, hi = ref.calculate_data_range(n_files)
    mean, std = ref.calculate_data_mean_std(n_files)
    paths = {f"{i}/{n}": ref.get_file_path(i, n)
             for n in (1, 13, 288, 289) for i in sorted({0, 1, 11, 12, n // 2, n - 1}) if i < n}
    with open(os.path.join(HERE, "data_golden.json"), "w") as fh:
        json.dump({"n_files": n_files, "seed0": 100, "shape": [1, 6, 7, 8],
                   "data_path": "/fake/wrfout", "vars": ref.DATA_VARS,
                   "scale": {k: list(v) for k, v in scale.items()},
                   "min": lo, "max": hi, "mean": mean, "std": std, "paths": paths}, fh, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "data":
        data_goldens()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "conv3d":
        conv3d_goldens()
        sys.exit(0)
    model_goldens()
    conv_goldens()
    regression_goldens()
    data_goldens()
    conv3d_goldens()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
