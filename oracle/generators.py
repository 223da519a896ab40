"""oracle/generators.py -- CPU restatement of the reference's regression row generator.

TEST INFRASTRUCTURE: only tests/ may import this (see oracle/__init__.py).  It restates, with plain
numpy slicing and the reference's own update rule, what `WeatherDataGenerator` of
/root/reference/multipleRegression.py yields, so that the device row builder
(wm_gather_cubes_f32) has an arbiter that is not product code:

  * gather_batches  (multipleRegression.py:236-257): for one file, the cube
    [r0:r0+c, r1:r1+c, r2:r2+c] of every tested variable's field `ds[var][0]`, stacked in
    DATA_VARS order -> (n_vars, c, c, c);
  * __getitem__     (:217-234): for the `batch` files start_index + index*batch ..., the cubes of
    file i (x1), file i + TIME_SPAN (x2) and file i + TIME_SPAN // 2 (y, flattened per row);
  * __call__        (:259-270): batch k uses index % length and the region
    (R0 % dim[1], R1 % dim[2], R2 % dim[3]); AFTERWARDS index += 1 and
    region_index = (index // length, OLD R0 // dim[1], OLD R1 // dim[2]) -- so the second and third
    coordinates lag one and two batches behind the first.

Parity unpinned by the reference itself (it holds no fixture and needs tensorflow to import); this
file follows the cited lines literally and the tests compare the kernel with it bit for bit.
"""
import numpy as np


def gather_batches(all_data, start_index, tested_vars, cube, index, region):
    """multipleRegression.py:236-257 with LOAD_DATA = True: all_data[index - start_index] maps a
    variable name to its (1, Z, Y, X) field."""
    ds = all_data[index - start_index]
    r0, r1, r2 = region
    parts = [np.asarray(ds[var][0][r0:r0 + cube, r1:r1 + cube, r2:r2 + cube]) for var in tested_vars]
    return np.stack(parts)


def get_item(all_data, start_index, tested_vars, cube, batch, time_span, index, region):
    """multipleRegression.py:217-234: one batch (x1, x2, y)."""
    start = int(start_index + index * batch)
    end = int(start_index + (index + 1) * batch)
    x1 = np.stack([gather_batches(all_data, start_index, tested_vars, cube, i, region)
                   for i in range(start, end)])
    x2 = np.stack([gather_batches(all_data, start_index, tested_vars, cube, i + time_span, region)
                   for i in range(start, end)])
    y = np.stack([gather_batches(all_data, start_index, tested_vars, cube, i + time_span // 2, region)
                  for i in range(start, end)])
    return x1, x2, np.reshape(y, (y.shape[0], -1))


def walk(all_data, start_index, tested_vars, cube, batch, time_span, length, data_dim, count,
         state=(0, (0, 0, 0))):
    """multipleRegression.py:259-270: the next `count` batches from generator state
    (index, region_index); returns (x1, x2, y) concatenated over the batches and the new state."""
    index, region_index = int(state[0]), tuple(int(v) for v in state[1])
    xs1, xs2, ys = [], [], []
    for _ in range(int(count)):
        region = (region_index[0] % data_dim[1], region_index[1] % data_dim[2],
                  region_index[2] % data_dim[3])
        x1, x2, y = get_item(all_data, start_index, tested_vars, cube, batch, time_span,
                             index % length, region)
        xs1.append(x1)
        xs2.append(x2)
        ys.append(y)
        index += 1
        region_index = (index // length, region_index[0] // data_dim[1],
                        region_index[1] // data_dim[2])
    return (np.concatenate(xs1), np.concatenate(xs2), np.concatenate(ys)), (index, region_index)
