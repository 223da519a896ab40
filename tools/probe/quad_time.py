"""gram_quad vs the previous route (WM_GRAM_QUAD=0 in a second process) at the quad shapes."""
import sys, time, os
import numpy as np
import torch
from weather_model_amd import ops
for F, P, n in ((16, 8, 100_000_000), (8, 4, 200_000_000), (16, 4, 100_000_000), (8, 8, 100_000_000)):
    X = ops.synth_uniform((n, F), seed=1)
    Y = ops.synth_uniform((n, P), seed=2)
    plan = ops.GramPlan(n, F, P, "cuda")
    for _ in range(3):
        plan.acc.zero_(); plan.accumulate(X, Y)
    torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        plan.acc.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.accumulate(X, Y); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    # float64 check on a slice
    m = 200_000
    Z = torch.cat([X[:m].double(), torch.ones((m, 1), device="cuda", dtype=torch.float64), Y[:m].double()], 1)
    ref = Z.T @ Z
    acc = torch.zeros_like(plan.acc); got = ops.gram(X[:m], Y[:m], acc)
    err = float(((got - ref).abs() / ref.abs().clamp_min(1e-30)).max())
    print(f"QUAD={os.environ.get('WM_GRAM_QUAD', '1')} F={F} P={P} n={n}: {ms:.3f} ms  {n / ms / 1e6:.1f} G rows/s  {n * 4 * (F + P) / ms / 1e9:.2f} TB/s  relerr {err:.2e}", flush=True)
    del X, Y, plan
    torch.cuda.empty_cache()
