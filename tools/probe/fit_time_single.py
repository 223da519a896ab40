"""Time the singleRegression drop-in's fit (F=432, P=216: 6^3 cubes) from a preloaded device generator."""
import cProfile, importlib, os, pstats, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import fakes
from weather_model_amd.dropin import singleRegression as sr
sr = importlib.reload(sr)
tmp = tempfile.mkdtemp()
n_files = 60
for k in range(n_files):
    open(os.path.join(tmp, f"f{k}"), "w").write("x")
sr.XTRAIN_DATA_PATH = tmp
files = {sr.getFilePath(i, (n_files,)): fakes.wrf_like_file(700 + i, ['T'], (1, 30, 120, 120), {'T': (-3.0, 5.0)})
         for i in range(n_files)}
sr.xr = fakes.FakeXarray(files)
sr.DS_DIM = (1, 30 // 6, 120 // 6, 120 // 6)
gen = sr.optimizeDataset()
gen.preload("cuda")
print("len", len(gen), "batch", gen.batchSize, "dsDim", gen.dsDim, flush=True)
model = sr.createCompiledModel((6, 6, 6))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
model.fit(gen, epochs=1, steps_per_epoch=256, verbose=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
h = model.fit(gen, epochs=3, steps_per_epoch=steps, verbose=0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
rows = steps * gen.batchSize
print(f"{dt * 1e3:.2f} ms per epoch of {rows} rows ({rows / dt / 1e6:.2f} M rows/s), ridge {model.ridge}, loss {h.history['loss']}", flush=True)
pr = cProfile.Profile(); pr.enable()
model.fit(gen, epochs=3, steps_per_epoch=steps, verbose=0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
