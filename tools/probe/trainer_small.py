import numpy as np, torch
from weather_model_amd import synth
from weather_model_amd.dropin import newConvolution as nc
T, Z, Y, X = 9, 1, 721, 1440
data = {v: torch.from_numpy(synth.uniform(100 + k, (T, 1, Z, Y, X))) for k, v in enumerate(nc.data_vars)}
shape = (T - 6, Z, Y, X)
ds = nc.CustomDataset(data, int(np.prod(shape)), shape, 0)
torch.manual_seed(3)
model = nc.CustomModel((2, 6)).cuda()
tr = nc.DeviceTrainer(model)
import sys
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tr.epoch(ds, bs, max_batches=30, shuffle=(len(sys.argv) < 3))
torch.cuda.synchronize()
