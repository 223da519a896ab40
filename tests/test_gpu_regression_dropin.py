"""GPU tests of the regression drop-ins: ``create_model(...).fit`` (Gram + normal equations in
libwm.so) against float64 numpy lstsq on the same rows.  Tolerance: coefficients 1e-4 relative
(SURVEY 8d), predictions 1e-4 absolute on O(1) data.  Parity vs the reference's own Keras training
is unpinned (tensorflow absent); the pin is the closed-form minimiser of the same loss."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fakes  # noqa: E402

pytestmark = pytest.mark.gpu


def _lstsq(X, Y):
    A = np.concatenate([X.astype(np.float64), np.ones((len(X), 1))], axis=1)
    return np.linalg.lstsq(A, Y.astype(np.float64), rcond=None)[0]


def _mr(n_files, shape):
    from weather_model_amd.dropin import multipleRegression as mr
    mr = importlib.reload(mr)
    mr.X_TRAIN_DATA_PATH = "/fake/wrfout"
    mr.ALL_FILES = [f"wrfout_{i:03d}" for i in range(n_files)]
    rng = np.random.default_rng(5)
    files = {}
    base = {v: rng.uniform(0.2, 0.8, size=shape).astype(np.float32) for v in mr.DATA_VARS}
    for i in range(n_files):
        arrays = {}
        for k, v in enumerate(mr.DATA_VARS):
            lo, hi = mr.DATA_VARS_MIN[v], mr.DATA_VARS_MAX[v]
            # smooth-in-time field + noise, mapped to the variable's physical range
            f = base[v] + 0.1 * np.sin(0.15 * i + k) + 0.02 * rng.standard_normal(shape)
            arrays[v] = (lo + (hi - lo) * f).astype(np.float32)
        files[mr.get_file_path(i)] = fakes.FakeDataset(arrays)
    mr.xr = fakes.FakeXarray(files)
    return mr


def test_multiple_regression_fit_matches_lstsq():
    n_files = 120
    mr = _mr(n_files, (1, 12, 14, 14))
    stats = [mr.DATA_VARS_MIN, mr.DATA_VARS_MAX, mr.DATA_VARS_MEAN, mr.DATA_VARS_STD]
    all_data = mr.load_data(n_files, 0, stats)          # the first 80 % of the files (:169)
    it = 0
    input_shape = mr.calculate_input_shape(it)
    data_dim = mr.calculate_data_shape(it)
    assert data_dim == (1, 2, 2, 2)
    model = mr.create_compiled_model(input_shape)

    # both generators index all_data relative to their own start_index (:238)
    start = int(n_files * mr.VALIDATION_SPLIT)
    hist, mod = mr.fit_model_with_generator(model, n_files, input_shape, data_dim, it,
                                            all_data, stats)
    assert mod is model
    assert hist.epoch == [0]
    assert set(hist.history) == {"loss", "mae", "msle", "val_loss", "val_mae", "val_msle"}

    # the same rows, gathered independently
    bs = mr.BATCH_SIZE[it]
    steps = int((((n_files * 0.8) // bs) - 1) * 8)
    length = int(((n_files * 0.8) // bs) - (mr.TIME_SPAN // bs) - 1)
    gen = mr.WeatherDataGenerator(n_files, input_shape, length, data_dim, it, all_data, start,
                                  stats)
    x1, x2, y = gen.rows(steps)
    X = np.concatenate([x1.reshape(len(x1), -1), x2.reshape(len(x2), -1)], axis=1)
    assert model.rows_seen == len(X) == steps * bs
    beta = _lstsq(X, y)
    K, c = model.coefficients
    got = np.concatenate([K, c[None]], axis=0)
    assert np.abs(got - beta).max() / np.abs(beta).max() < 1e-4
    resid = np.concatenate([X, np.ones((len(X), 1))], 1) @ beta - y
    assert hist.history["loss"][0] == pytest.approx(float((resid ** 2).mean()), rel=1e-3)
    assert hist.history["mae"][0] == pytest.approx(float(np.abs(resid).mean()), rel=1e-3)

    pred = model.predict([x1[:5], x2[:5]])
    assert pred.shape == (5, 1) and pred.dtype == np.float32
    ref = np.concatenate([X[:5], np.ones((5, 1))], 1) @ beta
    np.testing.assert_allclose(pred, ref, atol=1e-4)

    # predict_data: 6-tuple with the reference's quirks (min trackers stay 0)
    errs = mr.predict_data(model, input_shape, n_files, stats, it, mr.TIME_SPAN)
    assert len(errs) == 6 and errs[4] == 0 and errs[5] == 0 and errs[0] > 0

    # keras-ordered weights compose to the fitted map; round trip through set_weights and save
    w = model.get_weights()
    assert [a.shape for a in w] == [(2, 2), (2,), (2, 2), (2,), (2, 1), (1,)]
    m2 = mr.create_model(input_shape)
    m2.set_weights(w)
    np.testing.assert_allclose(m2.predict([x1[:5], x2[:5]]), pred, atol=1e-6)


def test_save_and_load_round_trip(tmp_path):
    from weather_model_amd.dropin._keras_like import LinearStackModel
    rng = np.random.default_rng(0)
    x1 = rng.uniform(size=(4096, 2, 1, 1, 1)).astype(np.float32)
    x2 = rng.uniform(size=(4096, 2, 1, 1, 1)).astype(np.float32)
    y = (0.3 * x1 + 0.6 * x2).reshape(4096, 2).astype(np.float32) + 0.1
    m = LinearStackModel((2, 1, 1, 1))
    m.compile(loss="mean_squared_error", metrics=["mae"])
    h = m.fit([x1, x2], y, verbose=0)
    assert h.history["loss"][0] < 1e-10 and h.history["mae"][0] < 1e-5
    K, c = m.coefficients
    np.testing.assert_allclose(K, np.kron(np.array([[0.3], [0.6]]), np.eye(2)), atol=1e-4)
    np.testing.assert_allclose(c, 0.1, atol=1e-4)
    m.save(str(tmp_path / "m"))
    m2 = LinearStackModel.load(str(tmp_path / "m.npz"))
    np.testing.assert_array_equal(m2.predict([x1[:3], x2[:3]]), m.predict([x1[:3], x2[:3]]))
    assert m.evaluate([x1, x2], y)[0] == pytest.approx(h.history["loss"][0], rel=1e-6, abs=1e-14)


def test_single_regression_p216_collinear_cubes():
    """singleRegression shape: 6^3 cubes, 432 -> 216, few rows and strongly collinear columns;
    the solve falls back to a small ridge and the fit still reproduces the training targets to
    the accuracy a min-norm float64 lstsq does."""
    from weather_model_amd.dropin import singleRegression as sr
    sr = importlib.reload(sr)
    rng = np.random.default_rng(3)
    n = 2000
    lat = rng.standard_normal((n, 40)).astype(np.float32)          # 40 latent factors
    mix = rng.standard_normal((40, 216)).astype(np.float32)
    f0 = lat @ mix
    f2 = (lat + 0.1 * rng.standard_normal((n, 40)).astype(np.float32)) @ mix
    y = 0.5 * (f0 + f2)
    x1 = f0.reshape(n, 6, 6, 6)
    x2 = f2.reshape(n, 6, 6, 6)
    model = sr.createCompiledModel((6, 6, 6))
    hist = model.fit((x1, x2, y), verbose=0)
    assert model.ridge > 0                                        # rank 80 < 432 columns
    assert set(hist.history) == {"loss", "mean_squared_error"}
    pred = model.predict([x1[:64], x2[:64]])
    assert pred.shape == (64, 216)
    scale = float(np.abs(y).max())
    assert np.abs(pred - y[:64]).max() / scale < 1e-3
    assert hist.history["loss"][0] / float((y ** 2).mean()) < 1e-6


def test_fit_streams_many_chunks():
    """Rows cross several staging chunks; the result equals a one-chunk fit up to the float32
    grouping of the partial sums (products summed in float32 over <= 1024 rows, then float64)."""
    from weather_model_amd.dropin._keras_like import LinearStackModel
    rng = np.random.default_rng(9)
    n = 50_000
    x1 = rng.uniform(size=(n, 3)).astype(np.float32)
    x2 = rng.uniform(size=(n, 3)).astype(np.float32)
    y = (x1 * 0.25 + x2 * 0.75 + 0.01 * rng.standard_normal((n, 3))).astype(np.float32)

    def batches():
        for s in range(0, n, 16):
            yield x1[s:s + 16], x2[s:s + 16], y[s:s + 16]
    a = LinearStackModel((3,))
    a.fit(batches, steps_per_epoch=n // 16, chunk_rows=4096, verbose=0)
    b = LinearStackModel((3,))
    b.fit((x1, x2, y), verbose=0)
    Ka, ca = a.coefficients
    Kb, cb = b.coefficients
    np.testing.assert_allclose(Ka, Kb, rtol=0, atol=5e-6)
    beta = _lstsq(np.concatenate([x1, x2], 1), y)
    assert np.abs(np.concatenate([Kb, cb[None]]) - beta).max() / np.abs(beta).max() < 1e-4
    assert torch.cuda.is_available()
