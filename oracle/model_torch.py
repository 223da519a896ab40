"""Eager torch-CPU, batch-1 restatement of the reference's conv interpolation model and its training
step.  TEST INFRASTRUCTURE ONLY: bench.py's `cpu_baseline` leg times it (the library path the
reference runs on a CPU: torch's aten conv2d / autograd / torch.optim.Adam, one grid point per step),
tests pin it against tests/golden/model_golden.npz (outputs, gradients and Adam steps of the
reference's own CustomModel).  The reference file itself cannot travel to the GPU box.

Follows /root/reference/newConvolution.py:
  layers                 :129-137   four nn.Conv2d (1->16 (2,1); 16->32, 32->32, 32->32 (3,1)),
                                    Linear(192,64), tanh, Linear(64,6)
  concatenate_bounds     :139-144   per channel, [start ; h_c ; end] along H (a Python loop over the
                                    channels with two concatenations each -- kept as a loop here,
                                    because that loop IS the reference's CPU cost)
  forward                :146-163   unbatched (C,H,W) input
  training_loop          :180-186   forward, MSE, backward, Adam step, zero_grad, per sample
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

PARAM_NAMES = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight",
               "conv3.bias", "conv4.weight", "conv4.bias", "linear1.weight", "linear1.bias",
               "linear2.weight", "linear2.bias")


class EagerModel(torch.nn.Module):
    def __init__(self, params):
        """params: name -> array/tensor with the reference's state_dict names and shapes."""
        super().__init__()
        self.p = torch.nn.ParameterDict({
            k.replace(".", "_"): torch.nn.Parameter(torch.as_tensor(params[k], dtype=torch.float32).clone())
            for k in PARAM_NAMES})

    def get(self, name):
        return self.p[name.replace(".", "_")]

    @staticmethod
    def _sandwich(h, first, last):
        # h: (C,1,6); first/last: (1,6) detached rows.  One channel at a time, as the reference.
        grown = None
        for c in range(h.shape[0]):
            piece = torch.cat((first, h[c]), dim=0)
            piece = torch.cat((piece, last), dim=0).unsqueeze(0)
            grown = piece if grown is None else torch.cat((grown, piece), dim=0)
        return grown

    def forward(self, x):
        """x: (1,2,6) -> (1,6)."""
        first = x[:, 0].clone().detach()
        last = x[:, 1].clone().detach()
        h = torch.relu(F.conv2d(x, self.get("conv1.weight"), self.get("conv1.bias")))
        for name in ("conv2", "conv3", "conv4"):
            h = self._sandwich(h, first, last)
            h = torch.relu(F.conv2d(h, self.get(name + ".weight"), self.get(name + ".bias")))
        v = torch.flatten(h)
        v = torch.tanh(F.linear(v, self.get("linear1.weight"), self.get("linear1.bias")))
        return F.linear(v, self.get("linear2.weight"), self.get("linear2.bias")).unsqueeze(0)


def train_steps(model, xs, ys, optimizer=None):
    """The reference's per-sample loop over xs (n,1,2,6), ys (n,1,6).  Returns the losses."""
    opt = optimizer or torch.optim.Adam(model.parameters())
    loss_fn = torch.nn.MSELoss()
    losses = []
    for x, y in zip(xs, ys):
        loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    return losses
