"""Per-kernel profile target: the blocked solve at F=432,P=216 and F=64,P=8 (run under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import ops
for F, P, n in [(432, 216, 20000), (64, 8, 20000)]:
    X = ops.synth_uniform((n, F), seed=1); Y = ops.synth_uniform((n, P), seed=2)
    plan = ops.GramPlan(n, F, P, "cuda")
    plan.accumulate(X, Y)
    for _ in range(5): plan.solve()
    torch.cuda.synchronize()
