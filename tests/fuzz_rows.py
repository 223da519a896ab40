"""Fuzz of the large-grid (row) conv kernels against the C oracle (test infrastructure; not
collected by pytest -- run by name on a GPU box): random shapes inside the staged weight-gradient
plan (>= 1.05 M outputs, <= 32 output channels, Cin*kD <= 45), forward, backward-data and
weight/bias gradient.  python tests/fuzz_rows.py [cases] [seed] [short]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from weather_model_amd import ops, synth
from oracle import cpu as ocpu
ocpu.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def relinf(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


worst = 0.0
for it in range(cases):
    kD = int(rng.choice([1, 1, 1, 2, 3])); kH = int(rng.integers(1, 5)); kW = int(rng.integers(1, 6))
    Cin = int(rng.integers(1, max(2, 45 // kD + 1))); Cout = int(rng.integers(1, 33))
    K = Cin * kD * kH * kW
    if K > 500:
        continue
    D = kD + int(rng.integers(0, 3)); W = int(rng.integers(68, 600)); B = int(rng.integers(1, 4))
    if rng.random() < 0.3:
        W = 64 * int(rng.integers(1, 5)) + kW - 1 + int(rng.choice([0, 0, 1, 63]))     # OW on and around 64 k
    W = max(W, 68)
    OW, OD = W - kW + 1, D - kD + 1
    if OW < 64:
        continue
    OH = int(np.ceil(1.06e6 / (B * OD * OW))); H = OH + kH - 1
    if len(sys.argv) > 3 and sys.argv[3] == "short":      # stripes of a few rows: a top every few units
        OH = int(rng.integers(1, 6)); H = OH + kH - 1
        B = int(np.ceil(1.06e6 / (OD * OH * OW)))
    flops = 2.0 * B * OD * OH * OW * Cout * K
    if flops > 12e9 or B * Cin * D * H * W > 3e7:
        continue
    shape = (B, Cin, D, H, W)
    x = synth.uniform(100 + it, shape, 0, 1); w = synth.uniform(200 + it, (Cout, Cin, kD, kH, kW), -1, 1)
    b = synth.uniform(300 + it, (Cout,), -1, 1); gy = synth.uniform(400 + it, (B, Cout, OD, OH, OW), 0, 1)
    t0 = time.time()
    xd, wd, bd, gd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, w, b, gy))
    e_f = relinf(ops.conv3d_forward(xd, wd, bd).cpu().numpy(), ocpu.conv3d_fwd(x, w, b))
    e_d = relinf(ops.conv3d_backward_data(gd, wd, shape).cpu().numpy(), ocpu.conv3d_bwd_data(gy, w, shape))
    gw, gb = ops.conv3d_backward_wb(xd, gd, w.shape)
    ow, ob = ocpu.conv3d_bwd_wb(x, gy, w.shape)
    e_w, e_b = relinf(gw.cpu().numpy(), ow), relinf(gb.cpu().numpy(), ob)
    worst = max(worst, e_f, e_d, e_w, e_b)
    print(f"case {it}: x{shape} k({kD},{kH},{kW}) Cout {Cout} K {K}: fwd {e_f:.1e} bwd_data {e_d:.1e} "
          f"gw {e_w:.1e} gb {e_b:.1e}  ({time.time() - t0:.1f} s)", flush=True)
    assert max(e_f, e_d, e_w, e_b) < 1e-5, "MISMATCH"
print("worst", worst)
