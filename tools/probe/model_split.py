"""Forward-only vs forward+backward time of the fused model kernel on one 721x1440 slice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import ops
B = 721 * 1440
p = (torch.rand(ops.MODEL_PARAM_COUNT, device="cuda") - 0.5) * 0.2
x = torch.rand(B, 2, 6, device="cuda")
y = torch.rand(B, 6, device="cuda")
ws = torch.empty((ops._lib.lib().wm_model_workspace_bytes(B),), device="cuda", dtype=torch.uint8)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps
print(f"forward only   {timeit(lambda: ops.model_forward(p, x)):.3f} ms")
print(f"forward+backward {timeit(lambda: ops.model_train(p, x, y=y, ws=ws)):.3f} ms", flush=True)
