"""Time LinearStackModel.fit fed by the singleRegression drop-in's device generator (6^3 cubes of one
variable, BATCH_SIZE 4: F = 432, P = 216) at scale, and profile the host side of an epoch."""
import cProfile
import pstats
import sys
import time

import torch

from weather_model_amd.dropin import singleRegression as sr

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
n_files, Z, Y, X = 2500, 12, 40, 40
gen = sr.WeatherGenerator("x", "y", "T", 4, 1, 0.2, (n_files, Z, Y, X), (1, Z - 5, Y - 5, X - 5), (0.0, 1.0))
gen._vol = torch.rand((n_files - gen.startIndex, 1, Z, Y, X), device="cuda")
assert gen.device_ready()
model = sr.createCompiledModel((6, 6, 6))
model.fit(gen, epochs=1, steps_per_epoch=1000, verbose=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
h = model.fit(gen, epochs=3, steps_per_epoch=steps, verbose=0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"rows/epoch {steps * 4}: {dt * 1e3:.2f} ms per epoch ({steps * 4 / dt / 1e6:.1f} M rows/s), "
      f"loss {h.history['loss'][-1]:.4e}, {h.params['rows_from']}", flush=True)
pr = cProfile.Profile()
pr.enable()
model.fit(gen, epochs=3, steps_per_epoch=steps, verbose=0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
