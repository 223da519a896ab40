"""Time LinearStackModel.fit fed by the multipleRegression drop-in's device generator at scale."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import fakes

def make(n_files, shape, tested):
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
            f = base[v] + 0.1 * np.sin(0.15 * i + k) + 0.02 * rng.standard_normal(shape).astype(np.float32)
            arrays[v] = (lo + (hi - lo) * f).astype(np.float32)
        files[mr.get_file_path(i)] = fakes.FakeDataset(arrays)
    mr.xr = fakes.FakeXarray(files)
    mr.VARIABLE_TEST[0] = {d: d in tested for d in mr.DATA_VARS}
    return mr

nvar = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
n_files = int(sys.argv[3]) if len(sys.argv) > 3 else 120
grid = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (12, 100, 100)
from weather_model_amd.dropin import multipleRegression as mr0
mr = make(n_files, (1,) + grid, list(mr0.DATA_VARS)[:nvar])
stats = [mr.DATA_VARS_MIN, mr.DATA_VARS_MAX, mr.DATA_VARS_MEAN, mr.DATA_VARS_STD]
all_data = mr.load_data(n_files, 0, stats)
it = 0
input_shape = mr.calculate_input_shape(it)
data_dim = mr.calculate_data_shape(it)
bs = mr.BATCH_SIZE[it]
length = int(((n_files * 0.8) // bs) - (mr.TIME_SPAN // bs) - 1)
start = int(n_files * mr.VALIDATION_SPLIT)
print("input_shape", input_shape, "batch", bs, "rows/epoch", steps * bs, "files", n_files, "rows per pass",
      length * bs, flush=True)
for chunk in (1 << 18, None, None):
    gen = mr.WeatherDataGenerator(n_files, input_shape, length, data_dim, it, all_data, start, stats)
    assert gen.device_ready()
    model = mr.create_compiled_model(input_shape)
    kw = {} if chunk is None else {"chunk_rows": chunk}
    model.fit(gen, epochs=1, steps_per_epoch=1024, verbose=0, **kw)          # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = model.fit(gen, epochs=3, steps_per_epoch=steps, verbose=0, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"chunk_rows={chunk}: {dt / 3 * 1e3:.2f} ms per epoch ({steps * bs / (dt / 3) / 1e9:.2f} G rows/s), "
          f"F={model.F} P={model.P} loss {h.history['loss'][-1]:.3e}", flush=True)
gen = mr.WeatherDataGenerator(n_files, input_shape, length, data_dim, it, all_data, start, stats)
model = mr.create_compiled_model(input_shape)
pr = cProfile.Profile()
pr.enable()
model.fit(gen, epochs=3, steps_per_epoch=steps, verbose=0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
