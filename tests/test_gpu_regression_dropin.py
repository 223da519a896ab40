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
    # epochs as in Keras: every epoch consumes the NEXT steps_per_epoch batches of the walking
    # generator; EarlyStopping(loss, patience=2) ends the run (multipleRegression.py:339-350)
    E = len(hist.epoch)
    assert hist.epoch == list(range(E)) and 3 <= E <= mr.EPOCHS[it]
    assert set(hist.history) == {"loss", "mae", "msle", "val_loss", "val_mae", "val_msle"}
    assert all(len(v) == E for v in hist.history.values())
    assert hist.params["solver"].startswith("closed-form") and "unpinned" in hist.params["parity"]
    assert hist.params["metrics_from"] == "rows"

    # the same rows, gathered independently ON THE HOST (fit took them from the device walk)
    bs = mr.BATCH_SIZE[it]
    steps = int((((n_files * 0.8) // bs) - 1) * 8)
    length = int(((n_files * 0.8) // bs) - (mr.TIME_SPAN // bs) - 1)
    gen = mr.WeatherDataGenerator(n_files, input_shape, length, data_dim, it, all_data, start,
                                  stats)
    x1, x2, y = gen.rows(steps * E)
    X = np.concatenate([x1.reshape(len(x1), -1), x2.reshape(len(x2), -1)], axis=1)
    assert model.rows_seen == len(X) == steps * bs * E
    beta = _lstsq(X, y)
    K, c = model.coefficients
    got = np.concatenate([K, c[None]], axis=0)
    assert np.abs(got - beta).max() / np.abs(beta).max() < 1e-4
    # the last epoch's loss / mae: the final map over the last epoch's rows
    last = slice((E - 1) * steps * bs, None)
    resid = np.concatenate([X[last], np.ones((steps * bs, 1))], 1) @ beta - y[last]
    assert hist.history["loss"][-1] == pytest.approx(float((resid ** 2).mean()), rel=1e-3)
    assert hist.history["mae"][-1] == pytest.approx(float(np.abs(resid).mean()), rel=1e-3)
    # the first epoch's loss: the fit over the first epoch's rows alone
    b0 = _lstsq(X[:steps * bs], y[:steps * bs])
    r0 = np.concatenate([X[:steps * bs], np.ones((steps * bs, 1))], 1) @ b0 - y[:steps * bs]
    assert hist.history["loss"][0] == pytest.approx(float((r0 ** 2).mean()), rel=1e-3)

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

    # evaluate() over a device generator: the same rows as the last epoch -> the last epoch's numbers
    gen_e = mr.WeatherDataGenerator(n_files, input_shape, length, data_dim, it, all_data, start, stats)
    gen_e._advance_by((E - 1) * steps)
    ev = model.evaluate(gen_e, steps=steps)
    assert ev[0] == pytest.approx(hist.history["loss"][-1], rel=1e-9)
    assert ev[1] == pytest.approx(hist.history["mae"][-1], rel=1e-9)
    assert ev[2] == pytest.approx(hist.history["msle"][-1], rel=1e-9)

    # the fit above never built a row (CUBE_SIZE 1: Gram and metrics read the resident volume); the
    # same fit over gathered rows (WM_FIT_WALK=0) gives the same history and the same map
    assert hist.params["rows_from"].startswith("volume")
    os.environ["WM_FIT_WALK"] = "0"
    try:
        m3 = mr.create_compiled_model(input_shape)
        h3, _ = mr.fit_model_with_generator(m3, n_files, input_shape, data_dim, it, all_data, stats)
    finally:
        del os.environ["WM_FIT_WALK"]
    assert h3.params["rows_from"].startswith("gathered")
    assert h3.epoch == hist.epoch
    for k, v in hist.history.items():
        np.testing.assert_allclose(h3.history[k], v, rtol=5e-6)
    K3, c3 = m3.coefficients
    # float32 products summed in another order, then a solve: 1e-6 of the O(1) coefficients
    np.testing.assert_allclose(K3, K, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(c3, c, rtol=1e-6, atol=1e-6)


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


def test_device_rows_bit_exact_vs_host_generator():
    """Row a9: wm_gather_cubes_f32's closed-form region walk against oracle/generators.py (the
    literal restatement of multipleRegression.py:217-270) AND the drop-in's host generator, bit for
    bit, over enough batches for all three region coordinates to move; then with 2^3 cubes of
    three variables."""
    from oracle import generators as ogen
    n_files = 120
    mr = _mr(n_files, (1, 12, 14, 14))
    stats = [mr.DATA_VARS_MIN, mr.DATA_VARS_MAX, mr.DATA_VARS_MEAN, mr.DATA_VARS_STD]
    all_data = mr.load_data(n_files, 0, stats)
    start = int(n_files * mr.VALIDATION_SPLIT)
    for it, cube, tested, bs in ((0, 1, ("T",), 16), (1, 2, ("T", "QVAPOR", "W"), 4)):
        mr.CUBE_SIZE = tuple(cube if k == it else 1 for k in range(mr.TOTAL_ITERATIONS))
        mr.BATCH_SIZE = tuple(bs if k == it else 16 for k in range(mr.TOTAL_ITERATIONS))
        mr.VARIABLE_TEST[it] = {d: d in tested for d in mr.DATA_VARS}
        input_shape = mr.calculate_input_shape(it)
        data_dim = mr.calculate_data_shape(it)                       # (1, 2, 2, 2)
        length = int(((n_files * 0.8) // bs) - (mr.TIME_SPAN // bs) - 1)
        mk = lambda: mr.WeatherDataGenerator(n_files, input_shape, length, data_dim, it,    # noqa: E731
                                             all_data, start, stats)
        host, dev = mk(), mk()
        assert dev.device_ready()
        # the resident volume is built ONCE, whichever way the device is named ("cuda" used to
        # compare unequal to the tensor's "cuda:0": every chunk re-stacked and re-uploaded all files,
        # 5.7 s per epoch of 4e6 rows instead of 0.8 ms)
        v0 = dev.device_volume()
        assert all(dev.device_volume(d) is v0 for d in ("cuda", "cuda:0", torch.device("cuda"),
                                                        torch.device("cuda", 0)))
        count = 9 * length + 3                   # region_index reaches (9, 4, 2): all three move
        x1, x2, y = host.rows(count)
        Xh = np.concatenate([x1.reshape(len(x1), -1), x2.reshape(len(x2), -1)], axis=1)
        # the arbiter: the oracle's walk over the same fields, from the same start state
        tested_in_order = [d for d in mr.DATA_VARS if d in tested]
        (o1, o2, oy), ostate = ogen.walk(all_data, start, tested_in_order, cube, bs, mr.TIME_SPAN,
                                         length, data_dim, count)
        Xo = np.concatenate([o1.reshape(len(o1), -1), o2.reshape(len(o2), -1)], axis=1)
        assert np.array_equal(Xo, Xh) and np.array_equal(oy, y)       # the host generator follows it
        # in three uneven pieces: the walk must resume from the saved state
        parts = [dev.device_rows(k) for k in (5, length + 1, count - length - 6)]
        Xd = torch.cat([p[0] for p in parts]).cpu().numpy()
        Yd = torch.cat([p[1] for p in parts]).cpu().numpy()
        assert Xd.shape == Xh.shape == (count * bs, 2 * len(tested) * cube ** 3)
        assert np.array_equal(Xd, Xo.astype(np.float32))
        assert np.array_equal(Yd, oy.astype(np.float32))
        assert dev.get_state() == host.get_state()
        assert tuple(dev.get_state()[1]) == tuple(ostate[1]) and dev.get_state()[0] == ostate[0]
        assert host.region_index[0] >= 9 and host.region_index[1] >= 1
        if cube == 1:
            # the same walk through device_walk (no rows built): the Gram of the oracle's rows
            dw = mk()
            assert dw.device_walk_ready("cuda", 2 * len(tested), len(tested))
            acc = torch.zeros((3 * len(tested) + 1,) * 2, device="cuda", dtype=torch.float64)
            for k in (5, length + 1, count - length - 6):
                ops_args = dw.device_walk(k)
                from weather_model_amd import ops
                ops.gram_walk(*ops_args, acc=acc)
            assert dw.get_state() == host.get_state()
            Zo = np.concatenate([Xo, np.ones((len(Xo), 1)), oy], axis=1).astype(np.float64)
            np.testing.assert_allclose(acc.cpu().numpy(), Zo.T @ Zo, rtol=2e-6)
        # a generator state beyond 2^31 batches: the kernel's 64-bit walk (the run above took
        # the 31-bit fast path)
        far = (2 ** 31 + 7, ((2 ** 31 + 7) // length, 2 ** 31 + 1, 5))
        host.set_state(far)
        dev.set_state(far)
        (o1, o2, oy), ostate = ogen.walk(all_data, start, tested_in_order, cube, bs, mr.TIME_SPAN,
                                         length, data_dim, length + 2, state=far)
        Xo = np.concatenate([o1.reshape(len(o1), -1), o2.reshape(len(o2), -1)], axis=1)
        host.rows(length + 2)
        Xd, Yd = dev.device_rows(length + 2)
        assert np.array_equal(Xd.cpu().numpy(), Xo.astype(np.float32))
        assert np.array_equal(Yd.cpu().numpy(), oy.astype(np.float32))
        assert dev.get_state() == host.get_state()
        assert tuple(dev.get_state()[1]) == tuple(ostate[1]) and dev.get_state()[0] == ostate[0]


def test_gather_cubes_explicit_origins_and_errors():
    """The random-cube form (convolution.py:118-147: origins drawn on the host) against numpy
    slicing, and the argument checks."""
    from weather_model_amd import ops, synth
    files, nv, Z, Y, X = 9, 2, 7, 11, 13
    vol = synth.uniform(31, (files, nv, Z, Y, X))
    rng = np.random.default_rng(4)
    rows, c = 37, (3, 4, 5)
    org = np.stack([rng.integers(0, files - 4, rows), rng.integers(0, Z - c[0] + 1, rows),
                    rng.integers(0, Y - c[1] + 1, rows), rng.integers(0, X - c[2] + 1, rows)],
                   axis=1).astype(np.int32)
    x1, x2, y = ops.gather_cubes(torch.from_numpy(vol).cuda(), c, 2, 4, rows, origins=org,
                                 separate=True)
    for r, (f, z0, y0, x0) in enumerate(org):
        cut = lambda ff: vol[ff, :, z0:z0 + c[0], y0:y0 + c[1], x0:x0 + c[2]].reshape(-1)   # noqa: E731
        assert np.array_equal(x1[r].cpu().numpy(), cut(f))
        assert np.array_equal(x2[r].cpu().numpy(), cut(f + 4))
        assert np.array_equal(y[r].cpu().numpy(), cut(f + 2))
    bad = org.copy()
    bad[3, 0] = files - 2                      # file + 4 leaves the volume
    with pytest.raises(ValueError, match="leaves the volume"):
        ops.gather_cubes(torch.from_numpy(vol).cuda(), c, 2, 4, rows, origins=bad)
    from weather_model_amd import _lib
    with pytest.raises(_lib.WmError, match="resident files"):
        ops.gather_cubes(torch.from_numpy(vol).cuda(), (1, 1, 1), 3, 6, 16,
                         walk={"batch": 16, "length": 4, "index": 0, "region": (0, 0, 0),
                               "dims": (2, 2, 2)})


def test_gather_cubes_point_major_view_matches_row_major():
    """wm_gather_cubes_strided_f32: the same logical volume stored point-major (Z, Y, X, files,
    vars) and passed as a strided view gives bit-identical rows -- the four-variables-per-thread
    path (cube 1, 8 variables), the scalar strided path (cube 2, 3 variables) and explicit origins."""
    from weather_model_amd import ops, synth
    for nv, cube, bs in ((8, 1, 16), (3, 2, 4), (4, 1, 5)):
        files, Z, Y, X = 40, 5, 9, 12
        host = synth.uniform(17 + nv, (files, nv, Z, Y, X))
        rowmajor = torch.from_numpy(host).cuda()
        stored = torch.from_numpy(np.ascontiguousarray(host.transpose(2, 3, 4, 0, 1))).cuda()
        view = stored.permute(3, 4, 0, 1, 2)
        assert view.shape == rowmajor.shape and not view.is_contiguous()
        length = (files - 6) // bs
        walk = {"batch": bs, "length": length, "index": 3, "region": (1, 0, 0),
                "dims": (Z - cube + 1, Y - cube + 1, X - cube + 1)}
        rows = bs * (7 * length + 2)
        a = ops.gather_cubes(rowmajor, (cube,) * 3, 3, 6, rows, walk=walk)
        b = ops.gather_cubes(view, (cube,) * 3, 3, 6, rows, walk=walk)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # against numpy slicing for a few rows of the first batches (state (3; 1,0,0), length)
        X0 = a[0].cpu().numpy()
        for r in (0, bs - 1, bs, 2 * bs + 1):
            bq = r // bs
            idx = 3 + bq
            f = (idx % length) * bs + r % bs
            R0 = 1 if bq == 0 else idx // length
            z0 = R0 % (Z - cube + 1)
            if bq <= 2:     # R1, R2 still come from the start state (0, 0) or from R0 = 1 // dim_z = 0
                cut = lambda ff: host[ff, :, z0:z0 + cube, 0:cube, 0:cube].reshape(-1)   # noqa: E731
                assert np.array_equal(X0[r], np.concatenate([cut(f), cut(f + 6)]))
        rng = np.random.default_rng(nv)
        org = np.stack([rng.integers(0, files - 6, 33), rng.integers(0, Z - cube + 1, 33),
                        rng.integers(0, Y - cube + 1, 33), rng.integers(0, X - cube + 1, 33)],
                       axis=1).astype(np.int32)
        a = ops.gather_cubes(rowmajor, (cube,) * 3, 3, 6, 33, origins=org, separate=True)
        b = ops.gather_cubes(view, (cube,) * 3, 3, 6, 33, origins=org, separate=True)
        assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_gram_and_metrics_straight_from_the_volume_match_the_gathered_rows():
    """wm_gram_walk_f32 / wm_linear_metrics_walk_f32 (CUBE_SIZE 1, 1..4 variables) against the
    gathered rows through wm_gram_f32 / wm_linear_metrics_f32: integer data -> the Gram is exact
    (every product and sum is an integer below 2^24 per float32 partial); random data -> 2e-6
    relative (float32 partial sums in another order); the metrics' predictions are the same bits per
    row, their squares are summed in float32 groups of up to 8 outputs before the float64 sums, and the
    groups differ (2e-6 relative).  Row-major and point-major volumes, walks that
    start mid-state, row counts that are not whole batches / trips, accumulation into a used acc."""
    from weather_model_amd import _lib, ops
    g = torch.Generator().manual_seed(3)
    for nv, bs, rows_extra, index, region in ((1, 16, 0, 0, (0, 0, 0)), (2, 5, 3, 3, (1, 0, 0)),
                                              (3, 7, 0, 11, (5, 2, 1)), (4, 16, 9, 2, (0, 1, 0))):
        files, Z, Y, X = 70, 3, 4, 5
        ints = torch.randint(0, 8, (files, nv, Z, Y, X), generator=g).float()
        rnd = torch.rand((files, nv, Z, Y, X), generator=g)
        length = (files - 6) // bs
        walk = {"batch": bs, "length": length, "index": index, "region": region, "dims": (Z, Y, X)}
        rows = bs * (9 * length + 1) + rows_extra
        for host, exact in ((ints, True), (rnd, False)):
            rowmajor = host.cuda()
            stored = host.permute(2, 3, 4, 0, 1).contiguous().cuda()
            view = stored.permute(3, 4, 0, 1, 2)
            Xr, Yr = ops.gather_cubes(rowmajor, (1, 1, 1), 3, 6, rows, walk=walk)
            want = ops.gram(Xr, Yr)
            for vol in (rowmajor, view):
                assert ops.walk_fused_supported(vol, (1, 1, 1))
                got = ops.gram_walk(vol, 3, 6, rows, walk)
                if exact:
                    assert torch.equal(got, want)
                else:
                    assert torch.allclose(got, want, rtol=2e-6, atol=0)
                again = ops.gram_walk(vol, 3, 6, rows, walk, acc=got.clone())
                assert torch.allclose(again, 2 * want, rtol=2e-6, atol=0)
            beta = torch.rand((2 * nv + 1, nv), generator=g, dtype=torch.float64).cuda() / (2 * nv)
            for msle in (True, False):
                ws = ops.linear_metrics(Xr, Yr, beta, want_msle=msle)
                for vol in (rowmajor, view):
                    gs = ops.linear_metrics_walk(vol, 3, 6, rows, walk, beta, want_msle=msle)
                    assert gs[3].item() == rows * nv
                    assert torch.allclose(gs, ws, rtol=2e-6, atol=0)
    # passes over the files longer than a wave's chunk of rows (256 or 512): the kernels then take one
    # cell per chunk, or two where a chunk crosses into the next pass, instead of decoding every row
    for nv, bs, index, region in ((1, 16, 0, (0, 0, 0)), (2, 5, 7, (1, 0, 0)), (3, 7, 200, (2, 1, 0)),
                                  (4, 16, 45, (1, 0, 0)), (1, 3, 1, (0, 0, 0))):
        files, Z, Y, X = 1400 // nv, 2, 3, 2
        host = torch.rand((files, nv, Z, Y, X), generator=g)
        length = (files - 6) // bs
        walk = {"batch": bs, "length": length, "index": index, "region": region, "dims": (Z, Y, X)}
        rows = bs * (13 * length + 5) + 3
        view = host.permute(2, 3, 4, 0, 1).contiguous().cuda().permute(3, 4, 0, 1, 2)
        Xr, Yr = ops.gather_cubes(view, (1, 1, 1), 3, 6, rows, walk=walk)
        beta = torch.rand((2 * nv + 1, nv), generator=g, dtype=torch.float64).cuda() / (2 * nv)
        for vol in (view, host.cuda()):
            assert torch.allclose(ops.gram_walk(vol, 3, 6, rows, walk), ops.gram(Xr, Yr), rtol=2e-6, atol=0)
            assert torch.allclose(ops.linear_metrics_walk(vol, 3, 6, rows, walk, beta),
                                  ops.linear_metrics(Xr, Yr, beta), rtol=2e-6, atol=0)
    # big enough for several trips of the whole grid, point-major
    files, nv, Z, Y, X = 2100, 1, 6, 10, 10
    host = torch.rand((files, nv, Z, Y, X), generator=g)
    view = host.permute(2, 3, 4, 0, 1).contiguous().cuda().permute(3, 4, 0, 1, 2)
    walk = {"batch": 64, "length": (files - 6) // 64, "index": 0, "region": (0, 0, 0), "dims": (Z, Y, X)}
    rows = 64 * walk["length"] * Z * Y * X - 17
    Xr, Yr = ops.gather_cubes(view, (1, 1, 1), 3, 6, rows, walk=walk)
    assert torch.allclose(ops.gram_walk(view, 3, 6, rows, walk), ops.gram(Xr, Yr), rtol=2e-6, atol=0)
    beta = torch.tensor([[0.3], [0.6], [0.05]], dtype=torch.float64).cuda()
    assert torch.allclose(ops.linear_metrics_walk(view, 3, 6, rows, walk, beta),
                          ops.linear_metrics(Xr, Yr, beta), rtol=2e-6, atol=0)
    # what it does not take
    assert not ops.walk_fused_supported(torch.zeros((8, 5, 2, 2, 2), device="cuda"), (1, 1, 1))
    assert not ops.walk_fused_supported(view, (2, 2, 2))
    with pytest.raises(_lib.WmError, match="1..4 variables"):
        ops.gram_walk(torch.zeros((80, 5, 2, 2, 2), device="cuda"), 3, 6, 16,
                      {"batch": 16, "length": 4, "index": 0, "region": (0, 0, 0), "dims": (2, 2, 2)})
    with pytest.raises(_lib.WmError, match="resident files"):
        ops.gram_walk(view[:60], 3, 6, 16,
                      {"batch": 16, "length": 4, "index": 0, "region": (0, 0, 0), "dims": (2, 2, 2)})
    assert ops.gram_walk(view, 3, 6, 0, walk).abs().sum().item() == 0
    # straight through the C ABI: a workspace that is too small is refused, nothing is launched
    import ctypes
    L = _lib.lib()
    acc = torch.zeros((4, 4), device="cuda", dtype=torch.float64)
    small_ws = torch.empty((64,), device="cuda", dtype=torch.uint8)
    strides = (ctypes.c_int64 * 5)(*view.stride())
    args = (view.data_ptr(), files, 1, Z, Y, X, strides, 3, 6, 1024, 64, walk["length"], 0, 0, 0, 0, Z, Y, X)
    assert L.wm_gram_walk_workspace_bytes(1) > 64 and L.wm_gram_walk_workspace_bytes(5) == 0
    rc = L.wm_gram_walk_f32(*args, acc.data_ptr(), small_ws.data_ptr(), small_ws.numel(), 0)
    assert rc < 0 and acc.abs().sum().item() == 0
    sums = torch.zeros((4,), device="cuda", dtype=torch.float64)
    rc = L.wm_linear_metrics_walk_f32(*args, beta.data_ptr(), 1, 1e-7, sums.data_ptr(), small_ws.data_ptr(),
                                      small_ws.numel(), 0)
    assert rc < 0 and sums.abs().sum().item() == 0
    with pytest.raises(_lib.WmError, match="leaves the grid"):
        ops.gram_walk(view, 3, 6, 16, dict(walk, dims=(Z + 1, Y, X)))


def test_gather_cubes_wave_per_row_kernel_matches_numpy_slicing():
    """Rows of 64..256 elements take gather_cubes_rows_kernel (a wave per row, singleRegression's 6^3
    cubes: p = 216): bit for bit numpy slicing along the reference's region walk (tuple update of
    singleRegression.py:330-340 restated below) and for explicit origins, at both ends of the size
    range, with several variables, on a row-major volume and a strided view, with a row count that
    leaves the last trip half empty."""
    from weather_model_amd import ops
    g = torch.Generator().manual_seed(9)
    for nv, cube, bs in ((1, (6, 6, 6), 4), (1, (4, 4, 4), 3), (3, (2, 4, 4), 5), (1, (4, 8, 8), 4),
                         (2, (5, 5, 5), 16)):
        files, Z, Y, X = 40, 8, 9, 10
        host = torch.rand((files, nv, Z, Y, X), generator=g)
        hn = host.numpy()
        p = nv * cube[0] * cube[1] * cube[2]
        assert 64 <= p <= 256
        length = (files - 2) // bs - 1
        dims = (Z - cube[0] + 1, Y - cube[1] + 1, X - cube[2] + 1)
        index, region = 2, (1, 0, 0)
        walk = {"batch": bs, "length": length, "index": index, "region": region, "dims": dims}
        rows = bs * (3 * length + 2) + 1
        want_x1 = np.empty((rows, p), np.float32)
        want_x2 = np.empty((rows, p), np.float32)
        want_y = np.empty((rows, p), np.float32)
        for r in range(rows):
            if r % bs == 0 and r > 0:
                index += 1
                region = (index // length, region[0] // dims[0], region[1] // dims[1])
            f = (index % length) * bs + r % bs
            z0, y0, x0 = region[0] % dims[0], region[1] % dims[1], region[2] % dims[2]
            cut = lambda ff: hn[ff, :, z0:z0 + cube[0], y0:y0 + cube[1], x0:x0 + cube[2]].reshape(-1)  # noqa: E731
            want_x1[r], want_x2[r], want_y[r] = cut(f), cut(f + 2), cut(f + 1)
        stored = host.permute(2, 3, 4, 0, 1).contiguous().cuda()
        for vol in (host.cuda(), stored.permute(3, 4, 0, 1, 2)):
            x1, x2, y = ops.gather_cubes(vol, cube, 1, 2, rows, walk=walk, separate=True)
            assert np.array_equal(x1.cpu().numpy(), want_x1)
            assert np.array_equal(x2.cpu().numpy(), want_x2)
            assert np.array_equal(y.cpu().numpy(), want_y)
            Xc, Yc = ops.gather_cubes(vol, cube, 1, 2, rows, walk=walk)
            assert np.array_equal(Xc.cpu().numpy(), np.concatenate([want_x1, want_x2], axis=1))
            assert np.array_equal(Yc.cpu().numpy(), want_y)
        rng = np.random.default_rng(nv + bs)
        org = np.stack([rng.integers(0, files - 2, 37), rng.integers(0, dims[0], 37),
                        rng.integers(0, dims[1], 37), rng.integers(0, dims[2], 37)], axis=1).astype(np.int32)
        x1, x2, y = ops.gather_cubes(host.cuda(), cube, 1, 2, 37, origins=org, separate=True)
        for r, (f, z0, y0, x0) in enumerate(org):
            cut = lambda ff: hn[ff, :, z0:z0 + cube[0], y0:y0 + cube[1], x0:x0 + cube[2]].reshape(-1)  # noqa: E731
            assert np.array_equal(x1[r].cpu().numpy(), cut(f))
            assert np.array_equal(x2[r].cpu().numpy(), cut(f + 2))
            assert np.array_equal(y[r].cpu().numpy(), cut(f + 1))


def test_gather_cubes_cube1_long_passes_match_the_reference_update_rule():
    """CUBE_SIZE 1 with passes over the files that are long against a wave's 64 units: the gather (and
    the volume-reading Gram / metrics kernels, which share cube_walk.h's chunk decode) against a
    literal host restatement of the reference's update rule (multipleRegression.py:259-270: every
    entry of the region index lags the one before it by a step), row by row, bit for bit -- scalar
    and four-variables-per-thread paths, row-major and point-major volumes, walks that start
    mid-state, several passes so that all three coordinates move."""
    from weather_model_amd import ops
    g = torch.Generator().manual_seed(21)
    for nv, bs, index, region in ((1, 16, 0, (0, 0, 0)), (4, 5, 7, (1, 0, 0)), (8, 16, 3, (2, 1, 0)),
                                  (3, 7, 130, (1, 1, 0)), (5, 4, 0, (0, 0, 0))):
        files, Z, Y, X = 500, 2, 2, 3
        host = torch.rand((files, nv, Z, Y, X), generator=g)
        hn = host.numpy()
        length = (files - 6) // bs
        dims = (Z, Y, X)
        walk = {"batch": bs, "length": length, "index": index, "region": region, "dims": dims}
        rows = bs * (7 * length + 3) + 2
        want_x = np.empty((rows, 2 * nv), np.float32)
        want_y = np.empty((rows, nv), np.float32)
        i, reg = index, region
        for r in range(rows):
            if r % bs == 0 and r > 0:
                i += 1
                reg = (i // length, reg[0] // dims[0], reg[1] // dims[1])
            f = (i % length) * bs + r % bs
            z0, y0, x0 = reg[0] % dims[0], reg[1] % dims[1], reg[2] % dims[2]
            want_x[r, :nv] = hn[f, :, z0, y0, x0]
            want_x[r, nv:] = hn[f + 6, :, z0, y0, x0]
            want_y[r] = hn[f + 3, :, z0, y0, x0]
        stored = host.permute(2, 3, 4, 0, 1).contiguous().cuda()
        for vol in (host.cuda(), stored.permute(3, 4, 0, 1, 2)):
            Xr, Yr = ops.gather_cubes(vol, (1, 1, 1), 3, 6, rows, walk=walk)
            assert np.array_equal(Xr.cpu().numpy(), want_x)
            assert np.array_equal(Yr.cpu().numpy(), want_y)
        if nv <= 4:
            Zh = np.concatenate([want_x, np.ones((rows, 1), np.float32), want_y], axis=1).astype(np.float64)
            got = ops.gram_walk(stored.permute(3, 4, 0, 1, 2), 3, 6, rows, walk).cpu().numpy()
            np.testing.assert_allclose(got, Zh.T @ Zh, rtol=2e-6)


def test_fit_without_resident_rows_reports_gram_loss():
    """Rows beyond max_resident_rows are not kept: the loss then comes from the Gram itself
    (MSE = (Y'Y - 2 b'Z'Y + b'Z'Zb) / nP) and agrees with the row-based loss."""
    from weather_model_amd.dropin._keras_like import LinearStackModel
    rng = np.random.default_rng(11)
    n = 40_000
    x1 = rng.uniform(size=(n, 2)).astype(np.float32)
    x2 = rng.uniform(size=(n, 2)).astype(np.float32)
    y = (0.4 * x1 + 0.5 * x2 + 0.05 * rng.standard_normal((n, 2))).astype(np.float32)

    def batches():
        for s in range(0, n, 16):
            yield x1[s:s + 16], x2[s:s + 16], y[s:s + 16]
    a = LinearStackModel((2,))
    a.compile(loss="mean_squared_error", metrics=["mae"])
    ha = a.fit(batches, steps_per_epoch=n // 16, chunk_rows=4096, verbose=0)
    b = LinearStackModel((2,))
    b.compile(loss="mean_squared_error", metrics=["mae"])
    hb = b.fit(batches, steps_per_epoch=n // 16, chunk_rows=4096, max_resident_rows=8192, verbose=0)
    assert ha.params["metrics_from"] == "rows" and hb.params["metrics_from"].startswith("gram")
    assert hb.history["loss"][0] == pytest.approx(ha.history["loss"][0], rel=1e-3)
    assert np.isnan(hb.history["mae"][0]) and ha.history["mae"][0] > 0
    np.testing.assert_array_equal(a.coefficients[0], b.coefficients[0])


def test_single_regression_preloaded_generator_bit_exact(tmp_path):
    """singleRegression.WeatherGenerator.preload(): 6^3 cubes of 'T' at files i, i+2; i+1 built on the
    GPU from one read of every file, against the host generator that opens three files per sample
    (singleRegression.py:251-299,330-340)."""
    from weather_model_amd.dropin import singleRegression as sr
    sr = importlib.reload(sr)
    n_files = 40
    for k in range(n_files):
        (tmp_path / f"f{k}").write_text("x")
    sr.XTRAIN_DATA_PATH = str(tmp_path)
    files = {sr.getFilePath(i, (n_files,)): fakes.wrf_like_file(
        700 + i, ['T'], (1, 15, 24, 24), {'T': (-3.0, 5.0)}) for i in range(n_files)}
    sr.xr = fakes.FakeXarray(files)
    assert sr.DS_DIM == (1, 3, 2, 2)
    host, dev = sr.optimizeDataset(), sr.optimizeDataset()
    dev.preload("cuda")
    assert dev.device_ready() and not host.device_ready()
    count = 3 * len(host) + 2
    gh = host()
    parts = [next(gh) for _ in range(count)]
    Xh = np.concatenate([np.concatenate([p[0].reshape(4, -1), p[1].reshape(4, -1)], axis=1)
                         for p in parts])
    Yh = np.concatenate([p[2] for p in parts])
    Xd, Yd = dev.device_rows(count)
    assert np.array_equal(Xd.cpu().numpy(), Xh.astype(np.float32))
    assert np.array_equal(Yd.cpu().numpy(), Yh.astype(np.float32))
    host._advance()                                  # the host generator advances after its yield
    assert dev.get_state() == host.get_state()


def test_single_regression_entry_point_fits_from_the_resident_volume(tmp_path):
    """singleRegression.fitModelWithGenerator() (:450-511) as the script calls it: the training files
    are read once (PRELOAD_ON_DEVICE) and every batch is built on the GPU; the same call with the
    reference's file access pattern gives the same history."""
    from weather_model_amd.dropin import singleRegression as sr
    sr = importlib.reload(sr)
    n_files = 40
    for k in range(n_files):
        (tmp_path / f"f{k}").write_text("x")
    sr.XTRAIN_DATA_PATH = str(tmp_path)
    files = {sr.getFilePath(i, (n_files,)): fakes.wrf_like_file(
        900 + i, ['T'], (1, 30, 60, 61), {'T': (-3.0, 5.0)}) for i in range(n_files)}
    sr.xr = fakes.FakeXarray(files)
    sr.EPOCHS = 2
    ha, ma = sr.fitModelWithGenerator()
    assert ha.params["rows_from"].startswith("gathered on the device")
    assert ma.rows_seen > 0 and np.isfinite(ha.history["loss"]).all()
    sr.PRELOAD_ON_DEVICE = False
    hb, mb = sr.fitModelWithGenerator()
    assert hb.params["rows_from"] == "host batches" and mb.rows_seen == ma.rows_seen
    # 56 rows against 432 features: an interpolating fit, the loss is rounding noise on targets of O(1)
    assert max(ha.history["loss"]) < 1e-9 and max(hb.history["loss"]) < 1e-9

