"""CPU tests of the ``convolution`` drop-in: names, generator shapes and values,
scaling, and that the model refuses to exist without a GPU.  Training/prediction parity is in
tests/test_gpu_conv3d_model.py."""
import importlib
import inspect
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fakes  # noqa: E402


def _conv(n_files=8, shape=(1, 24, 23, 22)):
    from weather_model_amd.dropin import convolution as cv
    cv = importlib.reload(cv)
    cv.DATA_PATH = "/fake/wrfout"
    cv.ALL_FILES = [f"wrfout_{i:02d}" for i in range(n_files)]
    lo, hi = cv.DATA_VARS_MIN['T'], cv.DATA_VARS_MAX['T']
    files = {f"{cv.DATA_PATH}/{name}": fakes.wrf_like_file(700 + i, ['T'], shape, {'T': (lo, hi)})
             for i, name in enumerate(cv.ALL_FILES)}
    cv.xr = fakes.FakeXarray(files)
    return cv


def test_convolution_surface_and_generator():
    cv = _conv()
    for name in ("CustomModel", "normalize", "append_to_array", "load_data", "DataGenerator",
                 "create_model", "compile_model", "get_prediction_data", "main", "INPUT_SHAPE",
                 "BATCH_SIZE", "DATA_VARIABLES", "DATA_VARS_MIN", "DATA_VARS_MAX"):
        assert hasattr(cv, name), name
    assert list(inspect.signature(cv.normalize).parameters) == ["var", "data"]
    assert list(inspect.signature(cv.get_prediction_data).parameters) == ["data", "var", "time"]
    data = cv.load_data()
    assert data['T'].shape == (8, 1, 24, 23, 22)
    assert 0.0 <= data['T'].min() and data['T'].max() <= 1.0
    raw = cv.xr.files["/fake/wrfout/wrfout_03"].arrays['T']
    np.testing.assert_allclose(
        data['T'][3], (raw - cv.DATA_VARS_MIN['T']) / (cv.DATA_VARS_MAX['T'] - cv.DATA_VARS_MIN['T']))
    gen = cv.DataGenerator(data)
    assert gen.data_shape == (4, 4, 3, 2) and len(gen) == 96
    random.seed(5)
    x1, x2, y = next(gen())
    assert x1.shape == (16, 1, 20, 20, 20) and x2.shape == x1.shape and y.shape == (16, 1, 8000)
    # first sample is index 0: corner cube at times 0, 4 and 2
    np.testing.assert_array_equal(x1[0, 0], data['T'][0][0][:20, :20, :20])
    np.testing.assert_array_equal(x2[0, 0], data['T'][4][0][:20, :20, :20])
    np.testing.assert_array_equal(y[0, 0], data['T'][2][0][:20, :20, :20].reshape(-1))
    np.testing.assert_array_equal(cv.get_prediction_data(data, 'T', 2), data['T'][2][0][:20, :20, :20])
    a = cv.append_to_array(None, np.zeros((2, 2)))
    assert cv.append_to_array(a, np.ones((2, 2))).shape == (2, 2, 2)


def test_convolution_model_needs_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cv = _conv()
    with pytest.raises(Exception):
        cv.create_model((1, 20, 20, 20))
