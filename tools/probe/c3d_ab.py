"""A/B of the Conv3D model's graphed training step in one process: class switches toggled per variant."""
import sys, time
import torch
from weather_model_amd.dropin._conv3d_model import Conv3DModel

def run(steps=400, **switches):
    for k, v in switches.items():
        setattr(Conv3DModel, k, v)
    m = Conv3DModel((1, 20, 20, 20), seed=0)
    m.compile()
    gen = torch.Generator(device="cuda").manual_seed(7)
    x1 = torch.rand((16, 1, 20, 20, 20), device="cuda", generator=gen)
    x2 = torch.rand((16, 1, 20, 20, 20), device="cuda", generator=gen)
    y = torch.rand((16, 1, 8000), device="cuda", generator=gen)
    for _ in range(20):
        m.graphed_train_step((x1, x2, y))
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        pending = None
        for _ in range(steps):
            issued = m.graphed_train_step((x1, x2, y), defer=True)
            if pending is not None:
                pending.result()
            pending = issued
        logs = pending.result()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3, logs["loss"]

variants = [dict(OVERLAP_ADAM=False), dict(OVERLAP_ADAM=True), dict(OVERLAP_ADAM=False), dict(OVERLAP_ADAM=True)]
if len(sys.argv) > 1:
    variants = [eval("dict(" + a + ")") for a in sys.argv[1:]]
for v in variants:
    ms, loss = run(**v)
    print(v, f"{ms:.4f} ms/step loss {loss:.6f}", flush=True)
