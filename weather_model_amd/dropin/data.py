"""Drop-in mirror of the reference's ``data.py``: dataset statistics helpers.

Same names and return values as /root/reference/data.py:28-100.  As in the reference the module
reads the global ``X_TRAIN_DATA_PATH`` (which the reference leaves commented out, :42 -- a caller
must set it); here it has the reference's commented value as default.  ``xarray`` is imported on
first use; tests assign the module attribute ``xr``.

The pooled standard deviation keeps the reference's arithmetic (:71,80,84): the per-variable sum of
``var * size`` is divided by the element count of ALL variables, so every std comes out too small
by sqrt(#variables) (0.30 instead of 1.0 for 11 unit-variance variables); and the mean is the mean
of per-file means (:78,83).  ``exact=True`` gives the per-variable pooled statistics instead.
"""
from __future__ import annotations

import fnmatch
import os

from . import _fields

DATA_VARS = ['T', 'P', 'PB', 'QVAPOR', 'QRAIN', 'QSNOW', 'QGRAUP', 'U', 'V', 'W', 'QCLOUD']  # :28-40
X_TRAIN_DATA_PATH = 'D:/Documents/Code/research/wrfout'

DEVICE_STATS = False    # True: per-file reductions run on the GPU (weather_model_amd.ops.field_stats)

xr = None


def _xr():
    global xr
    if xr is None:
        import xarray
        xr = xarray
    return xr


def calculate_data_range(data_size):
    """({var: min}, {var: max}) over files 0..data_size-1 (:52-66)."""
    max_value, min_value = {}, {}
    for i in range(data_size):
        ds = _xr().open_dataset(get_file_path(i, data_size))
        for data in DATA_VARS:
            hi, lo = _fields.summary(ds[data], DEVICE_STATS)[:2]
            if data not in max_value or hi > max_value[data]:
                max_value[data] = hi
            if data not in min_value or lo < min_value[data]:
                min_value[data] = lo
        ds.close()
    return min_value, max_value


def calculate_data_mean_std(data_size, exact=False):
    """({var: mean}, {var: std}) with the reference's pooling (:68-85); see the module docstring.
    ``exact=True`` (not in the reference) weights means by size and divides each variable's
    pooled second moment by its own element count, adding the between-file spread of the means."""
    mean_value, std_value = {}, {}
    sizes, wsum = {}, {}
    per_file = {}
    n = 0
    for i in range(data_size):
        ds = _xr().open_dataset(get_file_path(i, data_size))
        for data in DATA_VARS:
            _, _, m, s, size = _fields.summary(ds[data], DEVICE_STATS)
            mean_value[data] = mean_value.get(data, 0) + m
            std_value[data] = std_value.get(data, 0) + (s ** 2) * size
            sizes[data] = sizes.get(data, 0) + size
            wsum[data] = wsum.get(data, 0.0) + m * size
            per_file.setdefault(data, []).append((m, size))
            n += size
        ds.close()
    for data in DATA_VARS:
        if exact:
            mu = wsum[data] / sizes[data]
            between = sum(sz * (m - mu) ** 2 for m, sz in per_file[data])
            mean_value[data] = mu
            std_value[data] = ((std_value[data] + between) / sizes[data]) ** 0.5
        else:
            mean_value[data] = mean_value[data] / data_size
            std_value[data] = (std_value[data] / n) ** 0.5
    return mean_value, std_value


def get_file_path(index, data_size):
    """5-minute clock file name; the last index maps to day 21, hour 00 and keeps its minute
    (:88-96)."""
    last = index == (data_size - 1)
    minute = str((index % 12) * 5).zfill(2)
    hour = '00' if last else str(index // 12).zfill(2)
    return f"{X_TRAIN_DATA_PATH}/wrfout_d02_2023-06-2{'1' if last else '0'}_{hour}%3A{minute}%3A00"


def calculate_data_size():
    return len(fnmatch.filter(os.listdir(X_TRAIN_DATA_PATH), '*'))          # :98-100
