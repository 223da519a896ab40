"""Drop-in mirror of the reference's ``multipleRegression.py``.

Same module-level names, signatures, shapes and return values as
/root/reference/multipleRegression.py.  The Keras linear stack + Adam fit is replaced by the GPU
Gram/normal-equation path of libwm.so behind ``create_model(...).fit`` (see ``_keras_like``).

* ``xarray`` is imported on first use only (the reference imports it at the top, :5), so the
  module imports on a box without it; tests substitute the module attribute ``xr``.
* The reference's quirks are kept where its callers can see them, and named:
  ``calculate_data_mean_std`` divides the pooled variance by the element count of ALL variables
  (:406,418,424) and averages per-file means (:423); ``predict_data`` scores
  ``|prediction - average|`` (:456) and its min trackers never leave 0 (:445,462-465); the
  generator's region update rule (:268-270) is reproduced literally.
* ``WeatherDataGenerator.rows(count)`` is the vectorised form of ``count`` consecutive batches
  (one gather per variable instead of ``np.append`` growth, :134-138) -- same values, same order.
"""
from __future__ import annotations

import glob
import os
from datetime import datetime
from math import prod
from warnings import warn

import numpy as np

from . import _fields
from ._keras_like import EarlyStopping, History, LinearStackModel  # noqa: F401

X_TRAIN_DATA_PATH = 'D:/Documents/Code/research/wrfout'            # :12
KERAS_MODELS_PATH = 'D:/Documents/Code/research/keras-models'      # :13
DATA_OUT_PATH = 'D:/Documents/Code/research/output'                # :14
ALL_FILES = [os.path.basename(f) for f in glob.glob(X_TRAIN_DATA_PATH + "/wrfout*")]   # :15
DATA_VARS = ['T', 'P', 'PB', 'QVAPOR', 'QRAIN', 'QSNOW', 'U', 'V', 'W', 'QCLOUD']     # :16-27
TOTAL_ITERATIONS = 10
BATCH_SIZE = (16, ) * TOTAL_ITERATIONS
EPOCHS = (25,) * len(BATCH_SIZE)
VALIDATION_SPLIT = 0.2
SAVE_MODEL = (False, ) * len(BATCH_SIZE)
CALCULATE_STATS = True
NORMALIZE = (True,) * len(BATCH_SIZE)    # False implies standardization
CUBE_SIZE = (1, ) * len(BATCH_SIZE)
SHAPE_REDUCE = ([6, 7, 7],) * len(BATCH_SIZE)
LOAD_DATA = True
TIME_SPAN = 6                            # files skipped between x1 and x2; even (:38-40)

VARIABLE_TEST = [{data: data == d for data in DATA_VARS} for d in DATA_VARS]           # :43-45

# precalculated statistics of the WRF run (:48-94)
DATA_VARS_MIN = {'T': -8.417083740234375, 'P': -50.28515625, 'PB': 5420.28076171875,
                 'QVAPOR': 0.0, 'QRAIN': -1.3538986555136634e-12,
                 'QSNOW': -3.2677986886402296e-15, 'QGRAUP': 0.0, 'U': -14.949191093444824,
                 'V': -28.35787010192871, 'W': -5.611952781677246,
                 'QCLOUD': -7.121892919154105e-12}
DATA_VARS_MAX = {'T': 192.04437255859375, 'P': 1313.625, 'PB': 97532.5546875,
                 'QVAPOR': 0.02198491431772709, 'QRAIN': 0.0021312725730240345,
                 'QSNOW': 0.0001476061879657209, 'QGRAUP': 0.0, 'U': 26.568021774291992,
                 'V': 17.940385818481445, 'W': 8.062047958374023,
                 'QCLOUD': 0.002458590315654874}
DATA_VARS_MEAN = {'T': 35.421511191397805, 'P': 402.6549568572259, 'PB': 56756.6796875,
                  'QVAPOR': 0.005870880280334541, 'QRAIN': 5.818143079671262e-08,
                  'QSNOW': 9.50368343477367e-08, 'QGRAUP': 0.0, 'U': 1.3379405083128326,
                  'V': -0.38019444416422127, 'W': 0.002123257230844196,
                  'QCLOUD': 2.2856711307277937e-06}
DATA_VARS_STD = {'T': 12.86922902730058, 'P': 113.14531431220384, 'PB': 9284.244562274176,
                 'QVAPOR': 0.0018998037632266662, 'QRAIN': 3.8898476392190537e-07,
                 'QSNOW': 6.546815799049338e-07, 'QGRAUP': 0.0, 'U': 1.890336787350396,
                 'V': 2.3707455801620534, 'W': 0.041930900232612055,
                 'QCLOUD': 1.0485764700858156e-05}

DEVICE_STATS = False    # True: per-file reductions run on the GPU (ops.field_stats)

xr = None     # the xarray module, imported on first use (tests assign a stand-in)


def _xr():
    global xr
    if xr is None:
        import xarray
        xr = xarray
    return xr


class CustomModel(LinearStackModel):
    """Name kept for ``isinstance``/import parity with the reference's ``keras.Model`` subclass
    (:96-129); its ``train_step``/``test_step`` unpack ``(x1, x2, y)`` exactly as ``fit`` does."""


def append_to_array(array1, array2):
    """Grow along a new leading axis (:134-138)."""
    array2 = np.expand_dims(array2, axis=0)
    if array1 is None:
        return array2
    return np.append(array1, array2, axis=0)


def old_get_file_path(index, data_size):
    """File name from the 5-minute clock; the last index rolls over to the next day (:142-150)."""
    last = index == (data_size - 1)
    minute = str((index % 12) * 5).zfill(2)
    hour = '00' if last else str(index // 12).zfill(2)
    return f"{X_TRAIN_DATA_PATH}/wrfout_d02_2023-06-2{'1' if last else '0'}_{hour}%3A{minute}%3A00"


def get_file_path(index):
    return f"{X_TRAIN_DATA_PATH}/{ALL_FILES[index]}"                      # :152-153


def get_true_variables() -> set:
    """Variables enabled in at least one iteration (:157-163)."""
    return {data for iteration in range(TOTAL_ITERATIONS) for data in DATA_VARS
            if VARIABLE_TEST[iteration][data]}


def load_data(data_size, iteration, data_stats):
    """[{var: scaled ndarray}] for the training files (:167-187)."""
    all_data = []
    length = int(data_size * (1 - VALIDATION_SPLIT))
    true_vars = get_true_variables()
    for index in range(length):
        print(f"\rLoading data index: {index}/{length - 1}", end='')
        ds = _xr().open_dataset(get_file_path(index))
        scale = normalize if NORMALIZE[iteration] else standardize
        all_data.append({data: scale(ds[data], data, data_stats)
                         for data in DATA_VARS if data in true_vars})
        ds.close()
    print('\rLoading data index: Complete')
    return all_data


def normalize(data_array, data, data_stats) -> np.ndarray:
    """min-max scaling with data_stats = [min, max, mean, std] dicts (:190-192)."""
    lo, hi = data_stats[0][data], data_stats[1][data]
    return (data_array.to_numpy() - lo) / (hi - lo)


def standardize(data_array, data, data_stats) -> np.ndarray:
    return (data_array.to_numpy() - data_stats[2][data]) / data_stats[3][data]       # :195-197


class WeatherDataGenerator:
    """Yields ``(x1, x2, y)``: the tested variables' cube at files i, i+TIME_SPAN and
    i+TIME_SPAN/2, BATCH_SIZE consecutive i per batch (:200-270)."""

    def __init__(self, data_size, dim, length, data_dim, iteration, all_data, start_index,
                 data_stats):
        self.data_size = data_size
        self.dim = dim
        self.length = length
        self.data_dim = data_dim
        self.iteration = iteration
        self.index = 0
        self.region_index = (0, 0, 0)
        self.start_index = start_index
        self.data_stats = data_stats
        if LOAD_DATA:
            self.all_data = all_data

    def __len__(self):
        return self.length

    def __getitem__(self, index, region_index):
        bs = BATCH_SIZE[self.iteration]
        start = int(self.start_index + index * bs)
        files = range(start, start + bs)
        x1 = np.stack([self.gather_batches(i, region_index) for i in files])
        x2 = np.stack([self.gather_batches(i + TIME_SPAN, region_index) for i in files])
        y = np.stack([self.gather_batches(i + TIME_SPAN // 2, region_index) for i in files])
        return x1, x2, np.reshape(y, (y.shape[0], -1))

    def gather_batches(self, index, region_index):
        """(n_vars, c, c, c) cube of the tested variables of one file (:236-257)."""
        c = CUBE_SIZE[self.iteration]
        ds = self.all_data[index - self.start_index] if LOAD_DATA \
            else _xr().open_dataset(get_file_path(index))
        cubes = []
        for data in DATA_VARS:
            if not VARIABLE_TEST[self.iteration][data]:
                continue
            cube = ds[data][0][region_index[0]: region_index[0] + c,
                               region_index[1]: region_index[1] + c,
                               region_index[2]: region_index[2] + c]
            if not LOAD_DATA:
                scale = normalize if NORMALIZE[self.iteration] else standardize
                cube = scale(cube, data, self.data_stats)
            cubes.append(np.asarray(cube))
        if not LOAD_DATA:
            ds.close()
        return np.stack(cubes) if cubes else None

    def _region(self):
        return (self.region_index[0] % self.data_dim[1],
                self.region_index[1] % self.data_dim[2],
                self.region_index[2] % self.data_dim[3])

    def _advance(self):
        self.index += 1
        self.region_index = (self.index // self.__len__(),
                             self.region_index[0] // self.data_dim[1],
                             self.region_index[1] // self.data_dim[2])

    def _advance_by(self, count):
        """``count`` times ``_advance`` in closed form: after step t, region_index is
        (i_t // L, (i_{t-1} // L) // d1, ((i_{t-2} // L) // d1) // d2) -- every entry lags the one
        before it by a step, as the reference's tuple assignment has it (:259-270)."""
        count = int(count)
        if count < 3:
            for _ in range(count):
                self._advance()
            return
        i, L, d1, d2 = self.index + count, self.__len__(), self.data_dim[1], self.data_dim[2]
        self.index = i
        self.region_index = (i // L, ((i - 1) // L) // d1, (((i - 2) // L) // d1) // d2)

    def __call__(self):
        while True:                     # repeats over the data for ever (:259-270)
            yield self.__getitem__(self.index % self.__len__(), self._region())
            self._advance()

    def rows(self, count):
        """The next ``count`` batches as one ``(x1, x2, y)`` of ``count * BATCH_SIZE`` rows."""
        parts = []
        for _ in range(int(count)):
            parts.append(self.__getitem__(self.index % self.__len__(), self._region()))
            self._advance()
        return tuple(np.concatenate([p[k] for p in parts], axis=0) for k in range(3))

    # -------------------------------------------------------------- device-side rows (row a9)
    # Not in the reference: the same batches built on the GPU from a resident copy of all_data
    # (wm_gather_cubes_f32 evaluates the region walk above in closed form), so ``fit`` never sees a
    # host row.  ``LinearStackModel.fit`` uses this protocol when it finds it.
    def rows_per_batch(self):
        return BATCH_SIZE[self.iteration]

    def get_state(self):
        return self.index, tuple(self.region_index)

    def set_state(self, state):
        self.index, self.region_index = state[0], tuple(state[1])

    def device_ready(self):
        import torch
        return bool(LOAD_DATA and getattr(self, "all_data", None)) and torch.cuda.is_available()

    def device_volume(self, device="cuda"):
        """(files, n_tested_vars, Z, Y, X) float32 on the GPU, files from ``start_index`` on; built
        once per generator."""
        import torch
        vol = getattr(self, "_vol", None)
        want = torch.device(device)
        if want.type == "cuda" and want.index is None:        # "cuda" names the current device:
            want = torch.device("cuda", torch.cuda.current_device())   # compare like with like
        if vol is None or vol.device != want:
            tested = [d for d in DATA_VARS if VARIABLE_TEST[self.iteration][d]]
            # file by file: the host never holds a second copy of all_data
            rowmajor = None
            for k, f in enumerate(self.all_data):
                one = np.stack([np.asarray(f[d][0], dtype=np.float32) for d in tested])
                if rowmajor is None:
                    rowmajor = torch.empty((len(self.all_data),) + one.shape, device=want)
                rowmajor[k].copy_(torch.from_numpy(np.ascontiguousarray(one)))
            if CUBE_SIZE[self.iteration] == 1:
                # cube 1: a row takes ONE value per (file, variable) plane.  Stored point-major,
                # (Z, Y, X, files, vars), the rows of a batch (consecutive files at one grid point)
                # are one contiguous read; the kernel sees the same logical axes through strides
                # (the transposition runs on the device: numpy's strided copy of the same array
                # takes seconds per GB on the host)
                stored = rowmajor.permute(2, 3, 4, 0, 1).contiguous()
                vol = stored.permute(3, 4, 0, 1, 2)
            else:
                vol = rowmajor
            self._vol = vol
        return vol

    def device_rows(self, count, device="cuda"):
        """The next ``count`` batches as device tensors X (rows, 2p) = [x1 | x2], Y (rows, p) --
        bit for bit the values of ``rows(count)`` -- and the generator advanced past them."""
        from weather_model_amd import ops
        c = CUBE_SIZE[self.iteration]
        bs = BATCH_SIZE[self.iteration]
        vol = self.device_volume(device)
        walk = {"batch": bs, "length": self.__len__(), "index": self.index,
                "region": self.region_index, "dims": tuple(self.data_dim[1:4])}
        X, Y = ops.gather_cubes(vol, (c, c, c), TIME_SPAN // 2, TIME_SPAN, int(count) * bs,
                                walk=walk)
        self._advance_by(count)
        return X, Y

    def device_walk_ready(self, device, F, P):
        """May ``device_walk`` stand in for ``device_rows`` for a model of F features and P targets?
        CUBE_SIZE 1 (:35) with 1..4 tested variables: wm_gram_walk_f32 / wm_linear_metrics_walk_f32."""
        from weather_model_amd import ops
        c = CUBE_SIZE[self.iteration]
        vol = self.device_volume(device)
        return ops.walk_fused_supported(vol, (c, c, c)) and F == 2 * vol.shape[1] and \
            P == vol.shape[1] and self.index < (1 << 30)       # the kernels count batches in 31 bits

    def device_walk(self, count, device="cuda"):
        """The next ``count`` batches as a walk over the resident volume -- the arguments of
        ``ops.gram_walk`` / ``ops.linear_metrics_walk``: (vol, d_mid, d_end, rows, walk) -- and the
        generator advanced past them.  The rows are those of ``device_rows(count)``; no row is
        built."""
        bs = BATCH_SIZE[self.iteration]
        vol = self.device_volume(device)
        walk = {"batch": bs, "length": self.__len__(), "index": self.index,
                "region": self.region_index, "dims": tuple(self.data_dim[1:4])}
        self._advance_by(count)
        return vol, TIME_SPAN // 2, TIME_SPAN, int(count) * bs, walk


def create_model(input_shape):
    """start/end inputs -> Flatten -> Concatenate -> Dense(2p, linear) 'layer1' -> Dense(2p,
    linear) 'layer2' -> Dense(p) 'output' (:273-286), held as the affine map it spans."""
    return CustomModel(input_shape, hidden_layers=("layer1", "layer2"), output_name="output")


def create_compiled_model(input_shape):
    model = create_model(input_shape)
    model.compile(optimizer='adam', loss='mean_squared_error', weighted_metrics=['mae', 'msle'])
    return model                                                            # :289-298


def fit_model_with_generator(model, data_size, input_shape, data_dim, iteration, all_data,
                             data_stats):
    """(history, model) after fitting on the training generator, validated on the first
    VALIDATION_SPLIT of the files (:301-352).  Step/length arithmetic is the reference's."""
    bs, c = BATCH_SIZE[iteration], CUBE_SIZE[iteration]
    steps_per_epoch = int((((data_size * (1 - VALIDATION_SPLIT)) // bs) - 1)
                          * (data_dim[1] // c) * (data_dim[2] // c) * (data_dim[3] // c))
    length = int(((data_size * (1 - VALIDATION_SPLIT)) // bs) - (TIME_SPAN // bs) - 1)
    start_index = int(data_size * VALIDATION_SPLIT)
    dataset = WeatherDataGenerator(data_size, input_shape, length, data_dim, iteration, all_data,
                                   start_index, data_stats)
    length = int(((data_size * VALIDATION_SPLIT) // bs) - (TIME_SPAN // bs))
    validation_dataset = WeatherDataGenerator(data_size, input_shape, length, data_dim, iteration,
                                              all_data, 0, data_stats)
    callback = EarlyStopping(monitor='loss', patience=2)
    history = model.fit(
        dataset,
        epochs=EPOCHS[iteration],
        validation_data=validation_dataset if length > 0 else None,
        validation_steps=length,
        steps_per_epoch=steps_per_epoch,
        batch_size=bs,
        callbacks=[callback],
        verbose=1,
    )
    return history, model


def calculate_input_shape(iteration):
    n_vars = sum(1 for var in VARIABLE_TEST[iteration] if VARIABLE_TEST[iteration][var])
    return (n_vars,) + (CUBE_SIZE[iteration],) * 3                          # :355-361


def calculate_data_shape(iteration):
    ds = _xr().open_dataset(f'{X_TRAIN_DATA_PATH}/{ALL_FILES[0]}')
    shape = ds[DATA_VARS[0]].shape
    r = SHAPE_REDUCE[iteration]
    ds.close()
    return (shape[0], shape[1] // r[0], shape[2] // r[1], shape[3] // r[2])  # :364-373


def calculate_data_size():
    return len(glob.glob(X_TRAIN_DATA_PATH + "/wrfout*"))                   # :376-378


def calculate_data_range(data_size):
    """({var: min}, {var: max}) over the first data_size files (:381-399)."""
    print("Calculating min/max")
    max_value, min_value = {}, {}
    true_vars = get_true_variables()
    for i in range(data_size):
        ds = _xr().open_dataset(get_file_path(i))
        for data in DATA_VARS:
            if data not in true_vars:
                continue
            hi, lo = _fields.summary(ds[data], DEVICE_STATS)[:2]
            if data not in max_value or hi > max_value[data]:
                max_value[data] = hi
            if data not in min_value or lo < min_value[data]:
                min_value[data] = lo
        ds.close()
    return min_value, max_value


def calculate_data_mean_std(data_size):
    """({var: mean}, {var: std}) with the reference's pooling (:402-425): mean = mean of the
    per-file means; std = sqrt(sum_files var*size / n) where n counts the elements of EVERY
    variable, so the std is too small by sqrt(#variables).  Reproduced on purpose."""
    print("Calculating mean/std")
    mean_value, std_value = {}, {}
    n = 0
    true_vars = get_true_variables()
    for i in range(data_size):
        ds = _xr().open_dataset(get_file_path(i))
        for data in DATA_VARS:
            if data not in true_vars:
                continue
            _, _, m, sd, size = _fields.summary(ds[data], DEVICE_STATS)
            mean_value[data] = mean_value.get(data, 0) + m
            std_value[data] = std_value.get(data, 0) + (sd ** 2) * size
            n += size
        ds.close()
    for data in DATA_VARS:
        if data not in true_vars:
            continue
        mean_value[data] = mean_value[data] / data_size
        std_value[data] = (std_value[data] / n) ** 0.5
    return mean_value, std_value


def get_prediction_data(index, iteration, data_stats) -> np.ndarray:
    """Normalised corner cube of the tested variables of one file (:428-439)."""
    c = CUBE_SIZE[iteration]
    ds = _xr().open_dataset(get_file_path(index))
    data = None
    for d in DATA_VARS:
        if not VARIABLE_TEST[iteration][d]:
            continue
        data = append_to_array(data, normalize(ds[d][0][:c, :c, :c], d, data_stats))
    ds.close()
    return data


def predict_data(model, input_shape, data_size, data_stats, iteration, timespan):
    """6-tuple (sum avg err, sum pred err, max avg, max pred, min avg, min pred) over the first
    data_size//4 files (:442-471), including that the prediction is scored against the AVERAGE of
    the two inputs (:456) and that the min trackers start at 0 and so never move (:445,462-465)."""
    average_error, prediction_error = 0, 0
    average_max_error, prediction_max_error = 0, 0
    average_min_error, prediction_min_error = 0, 0
    for j in range(data_size // 4):
        data1 = np.expand_dims(get_prediction_data(j, iteration, data_stats), axis=0)
        data2 = np.expand_dims(get_prediction_data(j + timespan, iteration, data_stats), axis=0)
        actual_data = get_prediction_data(j + (timespan // 2), iteration, data_stats)
        prediction = model.predict([data1, data2]).reshape(input_shape)
        average = (data1 + data2) / 2
        avg_error = np.mean(abs(actual_data - average))
        pred_error = np.mean(abs(prediction - average))
        average_max_error = max(average_max_error, avg_error)
        prediction_max_error = max(prediction_max_error, pred_error)
        average_min_error = min(average_min_error, avg_error)
        prediction_min_error = min(prediction_min_error, pred_error)
        average_error += avg_error
        prediction_error += pred_error
    return (average_error, prediction_error, average_max_error,
            prediction_max_error, average_min_error, prediction_min_error)


def verify_parameters():
    """Sanity checks of the module configuration; False aborts ``main`` (:474-512)."""
    print(f"Running {TOTAL_ITERATIONS} iterations")
    if TOTAL_ITERATIONS <= 0:
        warn(f"Are you sure you want to run {TOTAL_ITERATIONS} iterations")
    for i in range(TOTAL_ITERATIONS):
        if EPOCHS[i] < 5:
            warn(f"Are you sure you want {EPOCHS} epochs?")
    for name, value in (("BATCH_SIZE", BATCH_SIZE), ("EPOCHS", EPOCHS), ("SAVE_MODEL", SAVE_MODEL),
                        ("NORMALIZE", NORMALIZE), ("CUBE_SIZE", CUBE_SIZE),
                        ("SHAPE_REDUCE", SHAPE_REDUCE), ("VARIABLE_TEST", VARIABLE_TEST)):
        if len(value) != TOTAL_ITERATIONS:
            warn(f"{name} does not match total iterations")
    if TIME_SPAN % 2 != 0:
        warn("TIME_SPAN is an odd number")
    if len(get_true_variables()) != len(DATA_VARS):
        warn("Some variables will not be tested")
    if not LOAD_DATA:
        print("Data will not be loaded")
    if len(ALL_FILES) == 0:
        warn("There are no files in TRAIN_DATA_PATH")
        return False
    ds = _xr().open_dataset(f'{X_TRAIN_DATA_PATH}/{ALL_FILES[0]}')
    for data in DATA_VARS:
        for shape in SHAPE_REDUCE:
            data_shape = ds[data].shape
            if (data_shape[1] // shape[0] < 1 or data_shape[2] // shape[1] < 1
                    or data_shape[3] // shape[2] < 1):
                warn(f"SHAPE_REDUCE is too small {data_shape} // {shape}")
                return False
    ds.close()
    if len(ALL_FILES) < 10:
        warn("Limited number of files")
        return False
    print("Note the following files must be in ascending order")
    print("If they are not then consider using old_get_file_path() over get_file_path()")
    for k in (0, len(ALL_FILES) // 4, len(ALL_FILES) // 2, int(len(ALL_FILES) // 1.25)):
        print(ALL_FILES[k])
    return True


def _report(file, tag, errs):
    avg, pred, avg_max, pred_max, avg_min, pred_min = errs
    file.write(f"Time span of {tag} \n")
    file.write(f"Average summed error: {np.sum(avg)} \n")
    file.write(f"Prediction summed error: {np.sum(pred)} \n")
    file.write(f"Average (max, min) error: {np.max(avg_max), np.min(avg_min)} \n")
    file.write(f"Prediction (max, min) error: {np.max(pred_max), np.min(pred_min)} \n")
    file.write(f"Average mean error: {np.mean(avg / 50)} \n")
    file.write(f"Prediction mean error: {np.mean(pred / 50)} \n")


def main():
    """The reference's experiment loop (:515-589): stats -> load -> per iteration fit, predict at
    TIME_SPAN and at 2, append a report to DATA_OUT_PATH/output.txt."""
    if not verify_parameters():
        return
    data_size = calculate_data_size()
    data_range = [DATA_VARS_MIN, DATA_VARS_MAX]
    data_mean_std = [DATA_VARS_MEAN, DATA_VARS_STD]
    if True in NORMALIZE and CALCULATE_STATS:
        data_range = calculate_data_range(data_size)
    if False in NORMALIZE and CALCULATE_STATS:
        data_mean_std = calculate_data_mean_std(data_size)
    data_stats = [data_range[0], data_range[1], data_mean_std[0], data_mean_std[1]]
    all_data = load_data(data_size, 0, data_stats) if LOAD_DATA else None

    for i in range(TOTAL_ITERATIONS):
        input_shape = calculate_input_shape(i)
        data_shape = calculate_data_shape(i)
        model = create_compiled_model(input_shape)
        hist, mod = fit_model_with_generator(model, data_size, input_shape, data_shape, i,
                                             all_data, data_stats)
        errs1 = predict_data(mod, input_shape, data_size, data_stats, i, TIME_SPAN)
        errs2 = predict_data(mod, input_shape, data_size, data_stats, i, 2)
        try:
            with open(f'{DATA_OUT_PATH}/output.txt', 'a+') as file:
                file.write(f"Ran model-{i} at {datetime.now()} \n")
                file.write(f"Batch size: {BATCH_SIZE[i]} | Cube size: {CUBE_SIZE[i]} | "
                           f"Variables tested: {VARIABLE_TEST[i]} | Shape reduction: "
                           f"{SHAPE_REDUCE[i]} | Normalization: {NORMALIZE} | Epochs Trained: "
                           f"{hist.epoch} | Time span: {TIME_SPAN} \n")
                file.write(f"{hist.history} \n")
                file.write(f"Looked through {data_size // 4} files \n")
                _report(file, TIME_SPAN, errs1)
                _report(file, 2, errs2)
                if SAVE_MODEL[i]:
                    file.write(f"Saving model to {KERAS_MODELS_PATH} \n")
                file.write("\n")
        except Exception as e:
            raise Exception(f'{i} - File could not be opened: {e}')
        if SAVE_MODEL[i]:
            os.mkdir(os.path.join(KERAS_MODELS_PATH, f"weather-model-{i}"))
            mod.export(f"{KERAS_MODELS_PATH}/weather-model-{i}")
        if i + 1 < TOTAL_ITERATIONS and NORMALIZE[i] != NORMALIZE[i + 1]:
            all_data = load_data(data_size, i + 1, data_stats)


if __name__ == '__main__':
    main()
