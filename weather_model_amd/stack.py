"""A stack of Conv2d(C -> C, kH x kW) + ReLU layers, forward and backward, on the HIP kernels.

BASELINE.json configs[3] read literally: "multi-layer 32-channel conv forecast model fwd+bwd on
0.25 deg grid, data-parallel with RCCL grad all-reduce".  The reference's own 32-channel layers
are ``nn.Conv2d(32, 32, ...)`` + ``nn.ReLU`` in sequence (newConvolution.py:131-133,152-159) with
``loss.backward()`` behind them (:184); this is that layer sequence at a grid size, without the
reference-specific (3,1) kernels and bounds glue (those live in csrc/model.hip).

forward : h_0 = x,  h_l = relu(conv(h_{l-1}, W_l) + b_l)          (ReLU fused into the conv store)
backward: g_L = dL/dh_L masked by h_L > 0, then for l = L..1
              (dW_l, db_l) = weight/bias gradient(h_{l-1}, g_l)
              g_{l-1}      = relu'(h_{l-1}) * conv^T(g_l, W_l)      (mask fused into the store)
All buffers are allocated once; one step launches 3L - 1 kernels (+ the finalize kernels) and the
gradients land in ONE flat buffer, ready for a single all-reduce (weather_model_amd.dp).
"""
from __future__ import annotations

import torch

from . import dp, ops


class ConvStack:
    def __init__(self, weights, biases, x_shape):
        if len(weights) != len(biases) or not weights:
            raise ValueError("one bias per weight, at least one layer")
        self.w = [ops._req(w, "weight") for w in weights]
        self.b = [ops._req(b, "bias") for b in biases]
        dev = self.w[0].device
        N, C, H, W = (int(v) for v in x_shape)
        self.shapes = [(N, C, H, W)]
        for w in self.w:
            Cout, Cin, kH, kW = w.shape
            n, c, h, ww = self.shapes[-1]
            if Cin != c or kH > h or kW > ww:
                raise ValueError(f"layer weight {tuple(w.shape)} does not fit input {self.shapes[-1]}")
            self.shapes.append((n, Cout, h - kH + 1, ww - kW + 1))
        L = len(self.w)
        self.h = [None] + [torch.empty(s, device=dev) for s in self.shapes[1:]]
        # gradient w.r.t. h_l for l = 1..L-1 (h_L's comes from the caller, h_0 needs none)
        self.g = [None] + [torch.empty(s, device=dev) for s in self.shapes[1:]]
        sizes = [w.numel() for w in self.w] + [b.numel() for b in self.b]
        self.flat_grads = torch.zeros((sum(sizes),), device=dev)
        views, off = [], 0
        for k in sizes:
            views.append(self.flat_grads[off:off + k])
            off += k
        self.gw = [v.view(w.shape) for v, w in zip(views[:L], self.w)]
        self.gb = views[L:]
        self.ws = [ops.conv2d_bwd_wb_workspace(self.shapes[l], self.w[l].shape, dev)
                   for l in range(L)]

    def forward(self, x):
        """h_L; every activation stays in ``self.h`` for the backward pass."""
        self.h[0] = ops._req(x, "input")
        for l, (w, b) in enumerate(zip(self.w, self.b)):
            ops.conv2d_forward(self.h[l], w, b, relu=True, out=self.h[l + 1])
        return self.h[-1]

    def backward(self, gout, premasked=False):
        """Gradients of every layer from dL/dh_L = ``gout`` (left untouched: its product with the
        mask h_L > 0 goes to the stack's own buffer unless ``premasked``).  Returns the flat
        gradient buffer [dW_1..dW_L, db_1..db_L]."""
        L = len(self.w)
        g = ops._req(gout, "grad_output")
        if not premasked:
            g = ops.relu_mask(g, self.h[L], out=self.g[L])
        for l in range(L, 0, -1):
            ops.conv2d_backward_wb(self.h[l - 1], g, self.w[l - 1].shape, ws=self.ws[l - 1],
                                   out_w=self.gw[l - 1], out_b=self.gb[l - 1])
            if l > 1:
                g = ops.conv2d_backward_data(g, self.w[l - 1], self.shapes[l - 1],
                                             relu_mask=self.h[l - 1], out=self.g[l - 1])
        return self.flat_grads

    def step_grads(self, x, gout):
        """forward + backward + (N > 1) the one gradient all-reduce of data parallelism."""
        self.forward(x)
        self.backward(gout)
        dp.reduce_grads_(self.flat_grads)
        return self.flat_grads
