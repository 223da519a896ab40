"""numpy float64 restatement of the reference's conv interpolation model.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/newConvolution.py:
  * CustomModel.__init__    :129-137  (conv1 1->16 (2,1); conv2 16->32 (3,1); conv3, conv4 32->32 (3,1);
                                        linear1 192->64; tanh; linear2 64->6)
  * concatenate_bounds      :139-144  (each conv input is [start ; h_c ; end] on the H axis, for
                                        every channel c; start/end are detached copies, :147-148)
  * forward                 :146-163
  * training_loop           :180-186  (MSELoss mean over the 6 outputs, backward, Adam step, zero_grad)
  * calculate_average       :166-172

The restatement is batched: ``x`` is (B, 2, 6) and every sample is treated exactly as the
reference treats its single (1, 2, 6) input.  Pinned against the reference itself by
tests/golden/model_golden.npz (written by tests/golden/make_golden.py, which imports the
reference's CustomModel).
"""
from __future__ import annotations

import numpy as np

PARAM_NAMES = (
    "conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
    "conv3.weight", "conv3.bias", "conv4.weight", "conv4.bias",
    "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
)
PARAM_SHAPES = {
    "conv1.weight": (16, 1, 2, 1), "conv1.bias": (16,),
    "conv2.weight": (32, 16, 3, 1), "conv2.bias": (32,),
    "conv3.weight": (32, 32, 3, 1), "conv3.bias": (32,),
    "conv4.weight": (32, 32, 3, 1), "conv4.bias": (32,),
    "linear1.weight": (64, 192), "linear1.bias": (64,),
    "linear2.weight": (6, 64), "linear2.bias": (6,),
}


def _f64(params):
    return {k: np.asarray(params[k], dtype=np.float64) for k in PARAM_NAMES}


def forward(params, x, return_cache=False):
    """x: (B,2,6) -> (B,6).  newConvolution.py:146-163."""
    p = _f64(params)
    x = np.asarray(x, dtype=np.float64).reshape(-1, 2, 6)
    s, e = x[:, 0, :], x[:, 1, :]                                  # (B,6) each  (:147-148)
    w1 = p["conv1.weight"][:, 0, :, 0]                             # (16,2)
    z1 = p["conv1.bias"][None, :, None] + w1[None, :, 0, None] * s[:, None, :] \
        + w1[None, :, 1, None] * e[:, None, :]                      # (B,16,6)   (:149)
    h = np.maximum(z1, 0.0)
    cache = {"s": s, "e": e, "z": [z1], "h": [h]}
    for name in ("conv2", "conv3", "conv4"):
        w = p[name + ".weight"][:, :, :, 0]                        # (O,C,3)
        a = w[:, :, 0].sum(axis=1)                                 # coefficient of start  (:139-144)
        c = w[:, :, 2].sum(axis=1)                                 # coefficient of end
        z = p[name + ".bias"][None, :, None] + a[None, :, None] * s[:, None, :] \
            + c[None, :, None] * e[:, None, :] + np.einsum("oc,bcw->bow", w[:, :, 1], h)
        h = np.maximum(z, 0.0)
        cache["z"].append(z)
        cache["h"].append(h)
    flat = h.reshape(h.shape[0], -1)                               # (B,192) order (chan,w)  (:160)
    u = flat @ p["linear1.weight"].T + p["linear1.bias"]
    t = np.tanh(u)
    out = t @ p["linear2.weight"].T + p["linear2.bias"]
    if return_cache:
        cache.update(flat=flat, t=t)
        return out, cache
    return out


def loss_and_grads(params, x, y):
    """Mean-squared error over all B*6 outputs (= mean of the reference's per-sample
    nn.MSELoss, :181,224) and its gradient w.r.t. the 12 forward-path parameters."""
    p = _f64(params)
    out, c = forward(p, x, return_cache=True)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 6)
    B = out.shape[0]
    diff = out - y
    per_sample = (diff ** 2).mean(axis=1)
    loss = per_sample.mean()
    g = {}
    dout = 2.0 * diff / (6.0 * B)
    g["linear2.weight"] = dout.T @ c["t"]
    g["linear2.bias"] = dout.sum(axis=0)
    du = (dout @ p["linear2.weight"]) * (1.0 - c["t"] ** 2)
    g["linear1.weight"] = du.T @ c["flat"]
    g["linear1.bias"] = du.sum(axis=0)
    dh = (du @ p["linear1.weight"]).reshape(B, 32, 6)
    s, e = c["s"], c["e"]
    for li, name in ((3, "conv4"), (2, "conv3"), (1, "conv2")):
        dz = dh * (c["z"][li] > 0.0)
        hin = c["h"][li - 1]
        w = p[name + ".weight"][:, :, :, 0]
        gw = np.zeros_like(p[name + ".weight"])
        gs = np.einsum("bow,bw->o", dz, s)
        ge = np.einsum("bow,bw->o", dz, e)
        gw[:, :, 0, 0] = gs[:, None]
        gw[:, :, 2, 0] = ge[:, None]
        gw[:, :, 1, 0] = np.einsum("bow,bcw->oc", dz, hin)
        g[name + ".weight"] = gw
        g[name + ".bias"] = dz.sum(axis=(0, 2))
        dh = np.einsum("oc,bow->bcw", w[:, :, 1], dz)
    dz1 = dh * (c["z"][0] > 0.0)
    gw1 = np.zeros_like(p["conv1.weight"])
    gw1[:, 0, 0, 0] = np.einsum("bow,bw->o", dz1, s)
    gw1[:, 0, 1, 0] = np.einsum("bow,bw->o", dz1, e)
    g["conv1.weight"] = gw1
    g["conv1.bias"] = dz1.sum(axis=(0, 2))
    return loss, per_sample, g


class Adam:
    """torch.optim.Adam defaults as the reference uses them (newConvolution.py:225):
    lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad."""

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = {k: np.zeros_like(np.asarray(params[k], dtype=np.float64)) for k in PARAM_NAMES}
        self.v = {k: np.zeros_like(np.asarray(params[k], dtype=np.float64)) for k in PARAM_NAMES}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        out = dict(params)
        for k in PARAM_NAMES:
            g = np.asarray(grads[k], dtype=np.float64)
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / np.sqrt(bc2) + self.eps
            out[k] = np.asarray(params[k], dtype=np.float64) - (self.lr / bc1) * self.m[k] / denom
        return out


def calculate_average(x):
    """newConvolution.py:166-172 -- the two-point temporal average baseline."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 2, 6)
    return (x[:, 0, :] + x[:, 1, :]) / 2.0
