"""GPU parity of the field kernels through the C ABI: statistics against a
float64 numpy restatement (1e-12 relative on sums; min/max/count exact), normalise bit-exact
against the reference's own ``normalize`` outputs (tests/golden/normalize_golden.npz) and numpy
float32 arithmetic, the sample gather bit-exact against the reference's ``CustomDataset``
(tests/golden/dataset_golden.npz) and numpy indexing."""
import os

import numpy as np
import pytest
import torch

from weather_model_amd import ops, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DATA_VARS = ['T', 'P', 'PB', 'QVAPOR', 'QRAIN', 'QSNOW']


@pytest.fixture(scope="module")
def ocpu():
    from oracle import cpu
    return cpu


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 1024, 65537, 3_000_001])
def test_field_stats_vs_numpy(n, ocpu):
    x = synth.uniform(40 + n % 7, (n,), -50.0, 1300.0)
    got = ops.field_stats(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = ocpu.field_stats(x)
    assert got[0] == ref[0] and got[1] == ref[1] and got[4] == ref[4] == n
    assert abs(got[2] - ref[2]) <= 1e-12 * abs(ref[3]) ** 0.5 * n ** 0.5 + 1e-9
    assert abs(got[3] - ref[3]) <= 1e-12 * ref[3]


def test_field_stats_running_merge_nan_and_reproducibility(ocpu):
    a = synth.uniform(51, (200_003,), -3.0, 5.0)
    b = synth.uniform(52, (77_777,), -9.0, 2.0)
    b[[5, 700, 7000]] = np.nan
    bd = torch.from_numpy(np.concatenate([b, np.zeros(3, np.float32)])[:len(b)].copy()).cuda()
    out = ops.field_stats(torch.from_numpy(a).cuda())
    out = ops.field_stats(bd, out)
    ref = ocpu.field_stats(np.concatenate([a, b]))
    got = out.cpu().numpy()
    assert got[0] == ref[0] and got[1] == ref[1] and got[4] == ref[4] == len(a) + len(b) - 3
    assert abs(got[2] - ref[2]) < 1e-7 and abs(got[3] - ref[3]) < 1e-9 * ref[3]
    again = ops.field_stats(bd, ops.field_stats(torch.from_numpy(a).cuda()))
    assert torch.equal(out, again)                       # fixed-order combine
    # mean / std as the drop-ins derive them
    mean, std = got[2] / got[4], (got[3] / got[4] - (got[2] / got[4]) ** 2) ** 0.5
    v = np.concatenate([a, b]).astype(np.float64)
    assert mean == pytest.approx(np.nanmean(v), rel=1e-12)
    assert std == pytest.approx(np.nanstd(v), rel=1e-10)


def test_normalize_bit_exact_vs_reference_golden(ocpu):
    g = np.load(os.path.join(HERE, "golden", "normalize_golden.npz"))
    from weather_model_amd.dropin import newConvolution as nc
    raw = torch.from_numpy(g["raw"]).cuda()
    for v in DATA_VARS:
        lo, hi = nc.DATA_VARS_MIN[v], nc.DATA_VARS_MAX[v]
        got = ops.affine_scale(raw, lo, hi - lo).cpu().numpy()
        np.testing.assert_array_equal(got, g[v])
    # large, ragged field against numpy float32 arithmetic
    x = synth.uniform(61, (1_000_003,), 5000.0, 98000.0)
    lo, hi = nc.DATA_VARS_MIN['PB'], nc.DATA_VARS_MAX['PB']
    got = ops.affine_scale(torch.from_numpy(x).cuda(), lo, hi - lo).cpu().numpy()
    np.testing.assert_array_equal(got, ocpu.affine_scale(x, lo, hi - lo))
    # standardize flavour, in place
    xd = torch.from_numpy(x).cuda()
    ops.affine_scale(xd, 56756.68, 9284.2446, out=xd)
    np.testing.assert_array_equal(xd.cpu().numpy(), ocpu.affine_scale(x, 56756.68, 9284.2446))


def test_gather_matches_reference_dataset_golden(ocpu):
    g = np.load(os.path.join(HERE, "golden", "dataset_golden.npz"))
    T, Z, Y, X = int(g["T"]), int(g["Z"]), int(g["Y"]), int(g["X"])
    # the fake volume make_golden.py fed to the reference's CustomDataset
    vol = np.stack([synth.uniform(100 + k, (T, 1, Z, Y, X))[:, 0] for k in range(6)], axis=-1)
    shape = (T - 6, Z, Y, X)
    vd = torch.from_numpy(np.ascontiguousarray(vol)).cuda()
    idx = torch.from_numpy(g["idxs"]).cuda()
    x, y = ops.gather_triplets(vd, shape, indices=idx)
    np.testing.assert_array_equal(x.cpu().numpy(), g["x"])
    np.testing.assert_array_equal(y.cpu().numpy(), g["y"])
    # contiguous range form == explicit indices; time offset; wrap-around and negative indices
    n = int(np.prod(shape))
    xr, yr = ops.gather_triplets(vd, shape, first=0, count=n)
    ox, oy = ocpu.gather_triplets(vol, shape, np.arange(n))
    np.testing.assert_array_equal(xr.cpu().numpy(), ox)
    np.testing.assert_array_equal(yr.cpu().numpy(), oy)
    odd = np.array([n + 3, 5 * n - 1, -1, -n], dtype=np.int64)
    xo, yo = ops.gather_triplets(vd, (2, Z, Y, X), indices=torch.from_numpy(odd).cuda(),
                                 time_offset=1)
    ox, oy = ocpu.gather_triplets(vol, (2, Z, Y, X), odd, time_offset=1)
    np.testing.assert_array_equal(xo.cpu().numpy(), ox)
    np.testing.assert_array_equal(yo.cpu().numpy(), oy)
    with pytest.raises(Exception):                       # time window leaves the volume
        ops.gather_triplets(vd, (T - 5, Z, Y, X), first=0, count=4)
    xe, ye = ops.gather_triplets(vd, shape, first=0, count=0)
    assert xe.shape == (0, 2, 6) and ye.shape == (0, 6)


def test_dataset_batch_uses_the_gather_kernel():
    from weather_model_amd.dropin import newConvolution as nc
    T, Z, Y, X = 9, 3, 4, 5
    data = {v: torch.from_numpy(synth.uniform(100 + k, (T, 1, Z, Y, X)))
            for k, v in enumerate(nc.data_vars)}
    shape = (T - 6, Z, Y, X)
    ds = nc.CustomDataset(data, int(np.prod(shape)), shape, 0)
    idx = [0, 1, 7, 19, 59, 60, 133, len(ds) - 1]
    xb, yb = ds.batch(idx, device="cuda")
    xs, ys = zip(*[ds[i] for i in idx])
    np.testing.assert_array_equal(xb.cpu().numpy(), torch.stack(xs).numpy())
    np.testing.assert_array_equal(yb.cpu().numpy(), torch.stack(ys).numpy())


def test_data_module_statistics_on_device_match_host_path():
    import importlib
    import json
    import sys
    sys.path.insert(0, HERE)
    import fakes
    from weather_model_amd.dropin import data
    data = importlib.reload(data)
    with open(os.path.join(HERE, "golden", "data_golden.json")) as fh:
        g = json.load(fh)
    data.X_TRAIN_DATA_PATH = g["data_path"]
    scale = {k: tuple(v) for k, v in g["scale"].items()}
    files = {data.get_file_path(i, g["n_files"]):
             fakes.wrf_like_file(g["seed0"] + i, g["vars"], tuple(g["shape"]), scale)
             for i in range(g["n_files"])}
    data.xr = fakes.FakeXarray(files)
    data.DEVICE_STATS = True
    lo, hi = data.calculate_data_range(g["n_files"])
    mean, std = data.calculate_data_mean_std(g["n_files"])
    for v in g["vars"]:
        assert lo[v] == g["min"][v] and hi[v] == g["max"][v]          # exact
        assert mean[v] == pytest.approx(g["mean"][v], rel=1e-5, abs=1e-6)
        assert std[v] == pytest.approx(g["std"][v], rel=1e-5)
