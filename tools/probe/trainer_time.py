"""DeviceTrainer.epoch (newConvolution drop-in) per batch, and the reference-shaped training_loop per step."""
import cProfile, pstats, time
import numpy as np
import torch
from weather_model_amd import synth
from weather_model_amd.dropin import newConvolution as nc

T, Z, Y, X = 9, 1, 721, 1440
data = {v: torch.from_numpy(synth.uniform(100 + k, (T, 1, Z, Y, X))) for k, v in enumerate(nc.data_vars)}
shape = (T - 6, Z, Y, X)
ds = nc.CustomDataset(data, int(np.prod(shape)), shape, 0)
torch.manual_seed(3)
model = nc.CustomModel((2, 6)).cuda()
tr = nc.DeviceTrainer(model)
for bs in (Y * X, 65536, 4096):
    nb = min(200, len(ds) // bs)
    tr.epoch(ds, bs, max_batches=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.epoch(ds, bs, max_batches=nb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nb
    print(f"DeviceTrainer batch {bs}: {dt * 1e3:.3f} ms per batch, {bs / dt / 1e6:.1f} M samples/s", flush=True)
pr = cProfile.Profile(); pr.enable()
tr.epoch(ds, 4096, max_batches=200)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
# the reference-shaped loop: batch of 64 samples per step through module + autograd + torch Adam
opt = torch.optim.Adam(model.parameters())
loss_fn = torch.nn.MSELoss()
loader = [ds.batch(torch.arange(k * 64, (k + 1) * 64), device="cuda") for k in range(50)]
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    nc.training_loop(loader[:5], model, loss_fn, opt, 50, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    nc.training_loop(loader, model, loss_fn, opt, 50, 0)
torch.cuda.synchronize()
print(f"training_loop (module + autograd + torch Adam), batch 64: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per step")
