import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import ops
torch.manual_seed(0)
for (N, Cin, H, W, Cout, kH, kW) in [(4, 32, 400, 802, 32, 3, 3), (4, 32, 400, 800, 32, 3, 3), (6, 16, 500, 401, 24, 3, 3), (5, 20, 240, 1000, 40, 2, 3), (8, 32, 721, 1440, 32, 3, 3)]:
    x = torch.rand(N, Cin, H, W, device="cuda")
    gy = torch.rand(N, Cout, H - kH + 1, W - kW + 1, device="cuda") - 0.5
    gw, gb = ops.conv3d_backward_wb(x.unsqueeze(2), gy.unsqueeze(2), (Cout, Cin, 1, kH, kW))
    xr = x.double().requires_grad_(False)
    wref = torch.zeros(Cout, Cin, kH, kW, device="cuda", dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xr, wref)
    (gwr,) = torch.autograd.grad(y, wref, gy.double())
    gbr = gy.double().sum((0, 2, 3))
    e1 = ((gw.squeeze(2).double() - gwr).abs().max() / gwr.abs().max()).item()
    e2 = ((gb.double() - gbr).abs().max() / gbr.abs().max()).item()
    print((N, Cin, H, W, Cout, kH, kW), f"gw err {e1:.2e} gb err {e2:.2e}", flush=True)
