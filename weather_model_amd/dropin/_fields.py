"""Per-file reductions for the statistics helpers of ``data`` / ``multipleRegression``.

``summary(da)`` returns (max, min, mean, std, size) of one xarray-like DataArray.  By default it
asks the array itself, as the reference does (``float(ds[v].max())`` ..., data.py:59-60,76-77);
with ``on_device=True`` the field is uploaded once and reduced by ``wm_field_stats_f32`` in one
pass (min, max, sum, sum of squares in float64 -- the mean/std then carry float64 accumulation
instead of numpy's float32 pairwise sums, a difference of ~1e-7 relative)."""
from __future__ import annotations

import numpy as np


def summary(da, on_device=False):
    if not on_device:
        return (float(da.max()), float(da.min()), float(da.mean()), float(da.std()), da.size)
    import torch

    from weather_model_amd import ops
    field = torch.from_numpy(np.ascontiguousarray(da.to_numpy(), dtype=np.float32)).cuda()
    mn, mx, s, ss, cnt = ops.field_stats(field).cpu().tolist()
    mean = s / cnt
    var = max(ss / cnt - mean * mean, 0.0)
    return (mx, mn, mean, var ** 0.5, da.size)
