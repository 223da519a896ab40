import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import ops
for (N, Cin, H, W, Cout, kH, kW) in [(4, 32, 400, 800, 32, 3, 3), (4, 32, 400, 834, 32, 3, 3)]:
    OH, OW = H - kH + 1, W - kW + 1
    x = torch.ones(N, Cin, H, W, device="cuda")
    gy = torch.ones(N, Cout, OH, OW, device="cuda")
    gw, gb = ops.conv3d_backward_wb(x.unsqueeze(2), gy.unsqueeze(2), (Cout, Cin, 1, kH, kW))
    exp = N * OH * OW
    print("ones: expected", exp, "gw min/max", gw.min().item(), gw.max().item(), "gb min/max", gb.min().item(), gb.max().item(), "per row-sample", (gb[0].item() - exp) / (N * OH))
    # column indicator: gy = 1 only at column c
    for c in (0, 63, 64, 767, 768, 770, OW - 1):
        gy.zero_(); gy[:, :, :, c] = 1
        gw, gb = ops.conv3d_backward_wb(x.unsqueeze(2), gy.unsqueeze(2), (Cout, Cin, 1, kH, kW))
        print("  col", c, "gb", gb[0].item() / (N * OH), "gw", gw[0, 0, 0, 0, 0].item() / (N * OH), gw[3, 5, 0, 2, 2].item() / (N * OH))
