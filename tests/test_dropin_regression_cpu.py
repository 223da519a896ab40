"""CPU tests of the regression-side drop-in modules: data, multipleRegression,
singleRegression, model -- names, shapes, generators, statistics quirks, no import side effects.
The GPU fit itself is covered by tests/test_gpu_regression_dropin.py."""
import importlib
import inspect
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fakes  # noqa: E402

from weather_model_amd.dropin import _keras_like  # noqa: E402


@pytest.fixture()
def golden_data():
    with open(os.path.join(HERE, "golden", "data_golden.json")) as fh:
        return json.load(fh)


# ------------------------------------------------------------------------------ data.py
def _data_module(g):
    from weather_model_amd.dropin import data
    data.X_TRAIN_DATA_PATH = g["data_path"]
    scale = {k: tuple(v) for k, v in g["scale"].items()}
    files = {data.get_file_path(i, g["n_files"]):
             fakes.wrf_like_file(g["seed0"] + i, g["vars"], tuple(g["shape"]), scale)
             for i in range(g["n_files"])}
    data.xr = fakes.FakeXarray(files)
    return data


def test_data_range_and_mean_std_match_reference(golden_data):
    """calculate_data_range / calculate_data_mean_std equal the reference's own outputs on the
    same fake files (tests/golden/make_golden.py data_goldens) -- pooled-std quirk included."""
    g = golden_data
    data = _data_module(g)
    assert data.DATA_VARS == g["vars"]
    lo, hi = data.calculate_data_range(g["n_files"])
    mean, std = data.calculate_data_mean_std(g["n_files"])
    for v in g["vars"]:
        assert lo[v] == g["min"][v] and hi[v] == g["max"][v]
        assert mean[v] == pytest.approx(g["mean"][v], rel=1e-12, abs=1e-15)
        assert std[v] == pytest.approx(g["std"][v], rel=1e-12)
    # the quirk: stds are too small by sqrt(#vars); exact=True removes it
    _, exact = data.calculate_data_mean_std(g["n_files"], exact=True)
    k = len(g["vars"]) ** 0.5
    for v in g["vars"]:
        a, b = g["scale"][v]
        true_std = (b - a) / 12 ** 0.5
        assert exact[v] == pytest.approx(true_std, rel=0.05)
        assert std[v] * k == pytest.approx(exact[v], rel=0.01)


def test_data_file_paths_match_reference(golden_data):
    data = _data_module(golden_data)
    for key, path in golden_data["paths"].items():
        i, n = (int(t) for t in key.split("/"))
        assert data.get_file_path(i, n) == path


def test_data_size_counts_directory_entries(tmp_path):
    from weather_model_amd.dropin import data
    for k in range(7):
        (tmp_path / f"wrfout_{k}").write_text("x")
    data.X_TRAIN_DATA_PATH = str(tmp_path)
    assert data.calculate_data_size() == 7


# ------------------------------------------------------------------------------ multipleRegression
def _mr(n_files=40, shape=(1, 12, 14, 21)):
    from weather_model_amd.dropin import multipleRegression as mr
    mr = importlib.reload(mr)
    mr.X_TRAIN_DATA_PATH = "/fake/wrfout"
    mr.ALL_FILES = [f"wrfout_{i:03d}" for i in range(n_files)]
    files = {mr.get_file_path(i): fakes.wrf_like_file(500 + i, mr.DATA_VARS, shape)
             for i in range(n_files)}
    mr.xr = fakes.FakeXarray(files)
    return mr


def test_multiple_regression_surface():
    mr = _mr()
    for name in ("append_to_array", "old_get_file_path", "get_file_path", "get_true_variables",
                 "load_data", "normalize", "standardize", "WeatherDataGenerator", "create_model",
                 "create_compiled_model", "fit_model_with_generator", "calculate_input_shape",
                 "calculate_data_shape", "calculate_data_size", "calculate_data_range",
                 "calculate_data_mean_std", "get_prediction_data", "predict_data",
                 "verify_parameters", "main", "CustomModel"):
        assert hasattr(mr, name), name
    assert list(inspect.signature(mr.WeatherDataGenerator.__init__).parameters)[1:] == [
        "data_size", "dim", "length", "data_dim", "iteration", "all_data", "start_index",
        "data_stats"]
    assert list(inspect.signature(mr.fit_model_with_generator).parameters) == [
        "model", "data_size", "input_shape", "data_dim", "iteration", "all_data", "data_stats"]
    assert list(inspect.signature(mr.predict_data).parameters) == [
        "model", "input_shape", "data_size", "data_stats", "iteration", "timespan"]
    assert mr.get_true_variables() == set(mr.DATA_VARS)
    assert mr.calculate_input_shape(3) == (1, 1, 1, 1)
    assert mr.calculate_data_shape(0) == (1, 2, 2, 3)
    assert mr.old_get_file_path(13, 289).endswith("2023-06-20_01%3A05%3A00")
    assert mr.old_get_file_path(288, 289).endswith("2023-06-21_00%3A00%3A00")
    a = mr.append_to_array(None, np.zeros((2, 3)))
    assert a.shape == (1, 2, 3) and mr.append_to_array(a, np.ones((2, 3))).shape == (2, 2, 3)


def test_multiple_regression_stats_and_scaling():
    mr = _mr(n_files=6)
    lo, hi = mr.calculate_data_range(6)
    mean, std = mr.calculate_data_mean_std(6)
    vals = np.stack([mr.xr.files[mr.get_file_path(i)].arrays['T'] for i in range(6)])
    assert lo['T'] == float(vals.min()) and hi['T'] == float(vals.max())
    assert mean['T'] == pytest.approx(np.mean([float(v.mean()) for v in vals]), rel=1e-12)
    # pooled over the elements of all 10 variables (the reference's n): too small by sqrt(10)
    pooled = (sum(float(v.std()) ** 2 * v.size for v in vals) / (vals.size * 10)) ** 0.5
    assert std['T'] == pytest.approx(pooled, rel=1e-12)
    stats = [lo, hi, mean, std]
    arr = fakes.FakeDataArray(vals[0])
    np.testing.assert_allclose(mr.normalize(arr, 'T', stats),
                               (vals[0] - lo['T']) / (hi['T'] - lo['T']))
    np.testing.assert_allclose(mr.standardize(arr, 'T', stats), (vals[0] - mean['T']) / std['T'])


def test_multiple_regression_generator_rows():
    """Batches are (x(t), x(t+6), x(t+3)) of the iteration's variable at one grid cell, 16
    consecutive files per batch; the region walks as the reference's update rule says; rows()
    is the same stream vectorised."""
    mr = _mr(n_files=64)
    data_size = 64
    stats = [mr.DATA_VARS_MIN, mr.DATA_VARS_MAX, mr.DATA_VARS_MEAN, mr.DATA_VARS_STD]
    all_data = mr.load_data(data_size, 0, stats)
    assert len(all_data) == int(64 * 0.8)
    it = 1                                             # variable 'P'
    data_dim = mr.calculate_data_shape(it)
    start = int(data_size * mr.VALIDATION_SPLIT)
    # all_data is indexed from the generator's start_index, as the reference does (:238)
    gen = mr.WeatherDataGenerator(data_size, (1, 1, 1, 1), 2, data_dim, it, all_data, 0, stats)
    stream = gen()
    x1, x2, y = next(stream)
    assert x1.shape == (16, 1, 1, 1, 1) and x2.shape == x1.shape and y.shape == (16, 1)
    for r in range(16):
        assert x1[r, 0, 0, 0, 0] == all_data[r]['P'][0][0, 0, 0]
        assert x2[r, 0, 0, 0, 0] == all_data[r + 6]['P'][0][0, 0, 0]
        assert y[r, 0] == all_data[r + 3]['P'][0][0, 0, 0]
    regions = [gen.region_index]
    for _ in range(5):
        next(stream)
        regions.append(gen.region_index)
    # index // len, then the previous region's components // data_dim (literal :268-270)
    assert regions[2] == (1, 0, 0) and regions[4] == (2, 0, 0)
    gen2 = mr.WeatherDataGenerator(data_size, (1, 1, 1, 1), 2, data_dim, it, all_data, 0, stats)
    rx1, rx2, ry = gen2.rows(3)
    gen3 = mr.WeatherDataGenerator(data_size, (1, 1, 1, 1), 2, data_dim, it, all_data, 0, stats)
    s3 = gen3()
    parts = [next(s3) for _ in range(3)]
    np.testing.assert_array_equal(rx1, np.concatenate([p[0] for p in parts]))
    np.testing.assert_array_equal(ry, np.concatenate([p[2] for p in parts]))
    assert start == 12


def test_oracle_walk_is_what_the_host_generator_yields():
    """oracle/generators.py (the literal restatement of multipleRegression.py:217-270 that arbitrates
    the device row builder) against the drop-in's generator, over enough batches for the second
    and third region coordinates to move (they lag one and two batches behind the first), and
    resumed from a saved state."""
    from oracle import generators as ogen
    mr = _mr(n_files=64, shape=(1, 3, 4, 5))
    stats = [mr.DATA_VARS_MIN, mr.DATA_VARS_MAX, mr.DATA_VARS_MEAN, mr.DATA_VARS_STD]
    all_data = mr.load_data(64, 0, stats)
    it = 1
    var = [d for d in mr.DATA_VARS if mr.VARIABLE_TEST[it][d]]
    data_dim = (1, 3, 4, 5)                              # every cell of the (3, 4, 5) field
    length = 2
    gen = mr.WeatherDataGenerator(64, (1, 1, 1, 1), length, data_dim, it, all_data, 0, stats)
    count = length * 3 * 4 * 2 + 5                       # R0 passes 3 * 4: R1 and R2 move
    x1, x2, y = gen.rows(count)
    (o1, o2, oy), state = ogen.walk(all_data, 0, var, 1, mr.BATCH_SIZE[it], mr.TIME_SPAN, length,
                                    data_dim, count)
    np.testing.assert_array_equal(o1, x1)
    np.testing.assert_array_equal(o2, x2)
    np.testing.assert_array_equal(oy, y)
    assert state == gen.get_state() and state[1][1] >= 1 and state[1][2] >= 1
    # literal check of the first batches against plain indexing: region (R0 % 3, R1 % 4, R2 % 5)
    b = mr.BATCH_SIZE[it]
    for k, reg in ((0, (0, 0, 0)), (2, (1, 0, 0)), (2 * 3, (0, 0, 0)), (2 * 3 + 1, (0, 1, 0))):
        r = k * b
        f = (k % length) * b
        assert o1[r, 0, 0, 0, 0] == all_data[f][var[0]][0][reg]
        assert o2[r, 0, 0, 0, 0] == all_data[f + mr.TIME_SPAN][var[0]][0][reg]
    (p1, _, _), st2 = ogen.walk(all_data, 0, var, 1, b, mr.TIME_SPAN, length, data_dim, 3, state=state)
    q1, _, _ = gen.rows(3)
    np.testing.assert_array_equal(p1, q1)
    assert st2 == gen.get_state()


def test_regression_models_need_the_gpu_library():
    """No CPU fallback: without a GPU the compute calls raise instead of answering."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    mr = _mr()
    model = mr.create_compiled_model((1, 1, 1, 1))
    assert model.metrics_names == ["loss", "mae", "msle"]
    x = np.zeros((4, 1, 1, 1, 1), np.float32)
    with pytest.raises(Exception):
        model.predict([x, x])
    with pytest.raises(Exception):
        model.fit((x, x, np.zeros((4, 1), np.float32)))


def test_keras_like_batch_sources():
    b = (np.zeros((2, 3)), np.ones((2, 3)), np.ones((2, 3)))
    assert len(list(_keras_like._batches(b, None))) == 1

    class Seq:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            return b
    assert len(list(_keras_like._batches(Seq(), None))) == 3
    assert len(list(_keras_like._batches(Seq(), 2))) == 2

    def gen():
        while True:
            yield b
    assert len(list(_keras_like._batches(gen, 5))) == 5
    assert len(list(_keras_like._batches(gen(), 4))) == 4
    with pytest.raises(ValueError):
        list(_keras_like._batches(gen, None))
    X, Y = _keras_like._as_rows(np.zeros((4, 1, 2, 2, 2)), np.ones((4, 1, 2, 2, 2)),
                                np.ones((4, 8)))
    assert X.shape == (4, 16) and Y.shape == (4, 8) and X.dtype == np.float32
    assert (X[:, :8] == 0).all() and (X[:, 8:] == 1).all()


# ------------------------------------------------------------------------------ singleRegression
def test_single_regression_import_touches_no_files(tmp_path):
    from weather_model_amd.dropin import singleRegression as sr
    sr = importlib.reload(sr)                         # would raise at import in the reference
    assert sr.xr is None and sr._lazy == {}
    dim = (289, 6, 6, 6)
    assert sr.getFilePath(0, dim).endswith("wrfout_d02_2023-06-20_00%3A00%3A00")
    assert sr.getFilePath(25, dim).endswith("2023-06-20_02%3A05%3A00")
    assert sr.getFilePath(288, dim).endswith("2023-06-21_00%3A00%3A00")
    # lazy globals
    n_files = 20
    for k in range(n_files):
        (tmp_path / f"f{k}").write_text("x")
    sr.XTRAIN_DATA_PATH = str(tmp_path)
    files = {sr.getFilePath(i, (n_files,)): fakes.wrf_like_file(
        900 + i, ['T'], (1, 30, 60, 60), {'T': (-3.0, 5.0)}) for i in range(n_files)}
    sr.xr = fakes.FakeXarray(files)
    assert sr.DIM == (n_files, 6, 6, 6)
    assert sr.DS_DIM == (1, 6, 6, 6)
    lo, hi = sr.DATA_RANGE
    assert lo < -2.9 and hi > 4.9
    gen = sr.optimizeDataset()
    x1, x2, y = next(gen())
    assert x1.shape == (4, 6, 6, 6) and x2.shape == (4, 6, 6, 6) and y.shape == (4, 216)
    start = int(n_files * 0.2)
    # NORMALIZE_DATA=True returns the raw field (the reference's inverted switch)
    np.testing.assert_array_equal(x1[0], files[sr.getFilePath(start, sr.DIM)].arrays['T'][0][:6, :6, :6])
    np.testing.assert_array_equal(x2[1], files[sr.getFilePath(start + 3, sr.DIM)].arrays['T'][0][:6, :6, :6])
    np.testing.assert_array_equal(y[2].reshape(6, 6, 6),
                                  files[sr.getFilePath(start + 3, sr.DIM)].arrays['T'][0][:6, :6, :6])
    v = sr.ValidationData(None, None, 'T', 0.2, 5, 4, sr.DIM)
    vx1, vx2, vy = v.getValidationData(0)
    assert vx1.shape == (4, 6, 6, 6) and vy.shape == (4, 216)
    m = sr.createCompiledModel((6, 6, 6))
    assert (m.F, m.P) == (432, 216) and m.layer_names == ("layer1", "output")


# ------------------------------------------------------------------------------ model.py
def test_model_module_has_no_import_side_effects_and_learns_xor(capsys):
    from weather_model_amd.dropin import model as toy
    toy = importlib.reload(toy)
    assert capsys.readouterr().out == ""              # the reference prints and trains at import
    x, y = toy.createDataset(50)
    assert x.shape == (51, 2) and y.shape == (51, 1)
    assert ((x[:, 0] ^ x[:, 1]) == y[:, 0]).all()
    m = toy.createCompiledModel()
    h = m.fit(x, y, batch_size=2, epochs=2, validation_split=0.2, verbose=0)
    assert h.epoch == [0, 1] and set(h.history) >= {"loss", "val_loss", "accuracy"}
    assert m.predict(np.array([[0, 0], [1, 0], [0, 1], [1, 1]])).shape == (4, 1)
    assert len(m.evaluate(x, y, batch_size=2)) == 3


def test_generator_advance_by_is_the_loop_in_closed_form():
    """``_advance_by(count)`` (what ``device_rows`` uses: a fit of 262 144 steps per epoch spent
    0.37 s per pass in the per-batch loop) against ``count`` calls of ``_advance``, whose tuple
    assignment makes every entry of the region index lag the one before it by a step
    (multipleRegression.py:259-270, singleRegression.py:302-312); from arbitrary states."""
    import random
    from weather_model_amd.dropin import multipleRegression as mr, singleRegression as sr

    random.seed(2)
    for cls, idx, reg in ((mr.WeatherDataGenerator, "index", "region_index"),
                          (sr.WeatherGenerator, "index", "regionIndex")):
        for _ in range(200):
            L, d = random.randint(1, 7), (1, random.randint(1, 4), random.randint(1, 4), 3)
            a, b = cls.__new__(cls), cls.__new__(cls)
            for g in (a, b):
                g.length = L
                g.data_dim = g.dsDim = d
            state = (random.randint(0, 60), tuple(random.randint(0, 9) for _ in range(3)))
            for g in (a, b):
                setattr(g, idx, state[0])
                setattr(g, reg, state[1])
            for count in (random.randint(0, 50) for _ in range(5)):
                for _ in range(count):
                    a._advance()
                b._advance_by(count)
                assert (getattr(a, idx), getattr(a, reg)) == (getattr(b, idx), getattr(b, reg))
