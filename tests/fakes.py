"""Stand-ins for the slice of xarray the reference's scripts use (open_dataset -> ds[var] ->
.to_numpy/.max/.min/.mean/.std/.size/.shape/indexing, ds.close()).  Test infrastructure only."""
import numpy as np

from weather_model_amd import synth


class FakeDataArray:
    def __init__(self, values):
        self.values = np.asarray(values)

    def to_numpy(self):
        return self.values

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __getitem__(self, key):
        return FakeDataArray(self.values[key])

    def __len__(self):
        return len(self.values)

    @property
    def shape(self):
        return self.values.shape

    @property
    def size(self):
        return self.values.size

    def max(self):
        return self.values.max()

    def min(self):
        return self.values.min()

    def mean(self):
        return self.values.mean()

    def std(self):
        return self.values.std()

    def __float__(self):
        return float(self.values)


class FakeDataset:
    def __init__(self, arrays):
        self.arrays = arrays
        self.closed = False

    def __getitem__(self, var):
        return FakeDataArray(self.arrays[var])

    def close(self):
        self.closed = True


class FakeXarray:
    """``open_dataset(path)`` served from a dict {path: FakeDataset}; records every open."""

    def __init__(self, files):
        self.files = files
        self.opened = []

    def open_dataset(self, path):
        self.opened.append(path)
        if path not in self.files:
            raise FileNotFoundError(path)
        return self.files[path]


def wrf_like_file(seed, variables, shape=(1, 6, 7, 8), scale=None):
    """One fake WRF output file: every variable a (1,Z,Y,X) float32 field, reproducible from
    (seed, variable position); ``scale[var] = (lo, hi)`` maps the U(0,1) draw to a range."""
    arrays = {}
    for k, var in enumerate(variables):
        a = synth.uniform(seed * 1000 + k, shape)
        lo, hi = (scale or {}).get(var, (0.0, 1.0))
        arrays[var] = (lo + (hi - lo) * a).astype(np.float32)
    return FakeDataset(arrays)
