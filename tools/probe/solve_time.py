"""Blocked solve wall time per variant library (WB_LIBS=comma list under tools/probe/variants)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("WB_LIB") is None and os.environ.get("WB_LIBS"):
    for lib in os.environ["WB_LIBS"].split(","):
        env = dict(os.environ, WB_LIB=lib)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
for F, P in [(432, 216), (64, 8)]:
    n = 20000
    X = ops.synth_uniform((n, F), seed=1); Y = ops.synth_uniform((n, P), seed=2)
    plan = ops.GramPlan(n, F, P, "cuda"); plan.accumulate(X, Y)
    for _ in range(3): plan.solve()
    torch.cuda.synchronize()
    res = []
    for r in range(5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(5): plan.solve()
        e[1].record(); torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]) / 5 * 1e3)
    print("%-22s F=%d P=%d solve %.1f us" % (os.environ.get("WB_LIB", "libwm.so"), F, P, np.median(res)), flush=True)
