#!/usr/bin/env python3
"""One training step of the Conv3D network (convolution.py shapes, batch 16) in a loop -- for
rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from weather_model_amd.dropin import convolution as cv
m = cv.compile_model(cv.create_model((1, 20, 20, 20)))
rng = np.random.default_rng(0)
x1 = torch.from_numpy(rng.uniform(size=(16, 1, 20, 20, 20)).astype(np.float32)).cuda()
x2 = torch.from_numpy(rng.uniform(size=(16, 1, 20, 20, 20)).astype(np.float32)).cuda()
y = torch.from_numpy(rng.uniform(size=(16, 1, 8000)).astype(np.float32)).cuda()
for i in range(3):
    m.train_step((x1, x2, y))
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for i in range(n):
    m.train_step((x1, x2, y))
torch.cuda.synchronize()
print(f"train_step {1e3 * (time.perf_counter() - t0) / n:.3f} ms (batch 16)")
for i in range(3):
    m.graphed_train_step((x1, x2, y))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    m.graphed_train_step((x1, x2, y))
torch.cuda.synchronize()
print(f"graphed_train_step {1e3 * (time.perf_counter() - t0) / n:.3f} ms (batch 16)")
