"""Shared case tables + input regeneration for the golden fixtures (see golden/make_golden.py)."""
import numpy as np

from weather_model_amd import synth

CONV_CASES = [
    ("ref_conv1", 3, 1, 2, 6, 16, 2, 1),
    ("ref_conv2", 3, 16, 3, 6, 32, 3, 1),
    ("ref_conv3", 3, 32, 3, 6, 32, 3, 1),
    ("seq_1x2", 2, 8, 1, 6, 16, 1, 2),
    ("k3x3_c1", 2, 1, 37, 53, 1, 3, 3),
    ("k3x3_c3to5", 1, 3, 19, 23, 5, 3, 3),
    ("k3x3_c32", 1, 32, 9, 11, 32, 3, 3),
    ("k5x5", 1, 2, 13, 17, 3, 5, 5),
    ("k7x7", 1, 1, 9, 11, 2, 7, 7),
    ("edge_1x1_out", 2, 2, 3, 3, 4, 3, 3),
    ("k1x1", 2, 4, 5, 7, 3, 1, 1),
    ("wide_row", 1, 1, 4, 1440, 1, 3, 3),
]

REG_CASES = [
    ("single_f1", 4096, 1, 1, 0, "uniform"),
    ("multi_p1", 4096, 2, 1, 21, "uniform"),
    ("f16", 65536, 16, 1, 1, "uniform"),
    ("f16_normal", 65536, 16, 1, 1, "normal"),
    ("f64", 32768, 64, 1, 7, "uniform"),
    ("f5_p3", 5000, 5, 3, 9, "uniform"),
    ("single_p216", 2048, 432, 216, 13, "normal"),
]


def conv_inputs(ci):
    name, N, Cin, H, W, Cout, kH, kW = CONV_CASES[ci]
    bnd = 1.0 / np.sqrt(Cin * kH * kW)
    x = synth.uniform(1000 + ci, (N, Cin, H, W))
    w = synth.uniform(2000 + ci, (Cout, Cin, kH, kW), -bnd, bnd)
    b = synth.uniform(3000 + ci, (Cout,), -bnd, bnd)
    gy = synth.uniform(4000 + ci, (N, Cout, H - kH + 1, W - kW + 1), -1.0, 1.0)
    return x, w, b, gy


def grid721_inputs(N=2):
    H, W = 721, 1440
    x = synth.uniform(2, (N, 1, H, W))
    w = synth.uniform(3, (1, 1, 3, 3), -1.0 / 3, 1.0 / 3)
    b = synth.uniform(5, (1,), -1.0 / 3, 1.0 / 3)
    gy = synth.uniform(4, (N, 1, H - 2, W - 2), -1.0, 1.0)
    return x, w, b, gy

# name, B, Cin, (D,H,W), Cout, (kD,kH,kW), seed, stride of the stored subsample
CONV3D_CASES = [
    ("small", 2, 3, (6, 7, 8), 5, (3, 2, 4), 31, 1),
    ("edge_one_output", 1, 2, (3, 4, 5), 17, (3, 4, 5), 32, 1),
    ("ragged_channels", 3, 33, (5, 5, 6), 18, (2, 3, 2), 33, 1),
    ("ref_layer1", 2, 1, (20, 20, 20), 32, (7, 7, 7), 34, 5),
    ("ref_layer2", 2, 16, (7, 7, 14), 64, (5, 5, 5), 35, 3),
]


# large 3-D grids: the persistent row kernels with a real depth axis (kD > 1, border rows in depth
# and height, rows that do not divide into 64-output segments, ragged channel tiles).  Checked
# against the oracle only (no committed vectors).
CONV3D_LARGE_CASES = [
    ("rows_fwd_bwd", 1, 8, (9, 70, 330), 24, (2, 3, 3), 36, 1),     # 185 k outputs
    ("rows_all", 2, 8, (20, 100, 330), 16, (2, 3, 3), 37, 1),       # 1.2 M outputs: also the LDS-staged weight gradient
    # three channel tiles x three tap tiles per wave: the weight gradient's float64 sums do not fit
    # in registers and take the slab path; depth of one (kD = 1 ring), shifted last segment
    ("rows_slab_sums", 3, 16, (1, 300, 1400), 40, (1, 3, 3), 44, 1),   # 1.25 M outputs
    ("rows_kh1", 2, 16, (1, 500, 1400), 16, (1, 1, 3), 45, 1),         # a ring of one slot: every unit refills
    # register-staged weight gradient: two channel tiles per wave with a ragged second one and a
    # ragged tap tile (K = 45); and the 2 x 2 wave split (seven tap tiles per wave, 30 channels)
    ("staged_ot2", 2, 5, (1, 600, 1000), 24, (1, 3, 3), 46, 1),         # 1.19 M outputs
    ("staged_w2", 1, 24, (1, 760, 1500), 30, (1, 3, 3), 47, 1),         # 1.14 M outputs
    # small volumes (plane-resident weight gradient): the bias column alone in the last tap tile
    # (K = 16), three channel tiles, planes of 80 positions; and a plane that is not a multiple of 4
    ("vol_bias_tile", 2, 4, (3, 9, 11), 40, (1, 2, 2), 38, 1),
    ("vol_odd_plane", 3, 5, (4, 6, 9), 7, (2, 2, 3), 39, 1),
    # mid-size outputs with a deep contraction (the reference's
    # second layer at batch 12; and two input-channel tiles with a ragged last one)
    ("mid_bwd_data", 12, 16, (7, 7, 14), 64, (5, 5, 5), 40, 1),
    ("mid_ct2", 3, 20, (8, 20, 20), 40, (3, 3, 3), 41, 1),
    # rows of <= 16 positions with mostly out-of-range taps: the row-tile backward-data kernel
    # (ragged channel tiles and channel groups, kW < 4; four channel tiles, kW = 6, full-width rows)
    ("rowtap_ragged", 3, 20, (5, 6, 11), 6, (4, 4, 3), 48, 1),
    ("rowtap_ct4", 2, 50, (6, 7, 16), 9, (4, 5, 6), 49, 1),
    ("rowtap_two_odd", 11, 16, (7, 7, 14), 20, (5, 5, 5), 50, 1),   # two batch items per workgroup, odd batch
    # >= 128 output planes with weights + slab resident in LDS: the plane-resident forward kernel
    ("vol_fwd", 10, 1, (20, 20, 20), 32, (7, 7, 7), 42, 1),
    ("vol_fwd_ragged", 9, 3, (18, 9, 13), 20, (3, 2, 4), 43, 1),
]


def conv3d_inputs(case):
    name, B, Cin, dhw, Cout, k, seed, _ = case
    x = synth.uniform(seed * 10 + 1, (B, Cin) + dhw)
    bound = 1.0 / np.sqrt(Cin * k[0] * k[1] * k[2])
    w = synth.uniform(seed * 10 + 2, (Cout, Cin) + k, -bound, bound)
    b = synth.uniform(seed * 10 + 3, (Cout,), -bound, bound)
    out = tuple(d - kk + 1 for d, kk in zip(dhw, k))
    gy = synth.uniform(seed * 10 + 4, (B, Cout) + out, -1.0, 1.0)
    return x, w, b, gy
