"""The Conv3D interpolation network of the reference's ``convolution.py`` (:156-180) on the GPU
operators of libwm.so, behind the slice of the ``keras.Model`` surface that script uses.

Per branch (two branches with their own weights, inputs ``start`` and ``end``):
  Conv3D(32, 7, channels_first) -> MaxPool3D(2, 2) -> Conv3D(64, 5, channels_first) ->
  MaxPool3D(2, 2) -> Flatten; then Concatenate -> Dense(relu) -> Dense(prod(input_shape)).
The pools carry no ``data_format`` in the reference, so Keras pools in its default channels_last
layout: on the (B,32,14,14,14) conv output the window runs over the (32,14,14) axes and the last
axis is left alone -> (B,16,7,7,14); the second conv then sees 16 input channels.
That is reproduced (``pool_channels_last=True``); ``False`` gives the presumably intended pool.

Convolutions, pools and their gradients are the hand-written kernels (``ops.conv3d``,
``ops.maxpool3d``), loss + metrics + loss gradient are one hand-written pass (``ops.mse_metrics``);
the two Dense layers are the skinny MFMA products of ``dense.hip`` (``ops.dense``: bias + ReLU in the
forward store, dense1's ReLU backward in the output layer's input-gradient store; widths that are
not multiples of 4 floats are refused at construction -- the reference's 640 and 8000 are) and Adam
is ``torch.optim.Adam`` with Keras' epsilon (its fused multi-tensor kernel moves the 5.8 M
parameters' 162 MB in 44 us, i.e. at the memory rate: a hand-written one would not be faster).
No CPU path.

Parity: unpinned against the reference's own run (tensorflow/keras absent here); the operators are
pinned against the float64 oracle and torch-CPU vectors, the assembled network against a plain
torch restatement on CPU (tests/test_gpu_conv3d_model.py).
"""
from __future__ import annotations

from math import prod

import numpy as np
import torch

from weather_model_amd import ops
from ._keras_like import History, _MSLE_EPS, _batches


def _glorot(shape, fan_in, fan_out, gen, device):
    bound = (6.0 / (fan_in + fan_out)) ** 0.5          # keras glorot_uniform
    t = torch.empty(shape, dtype=torch.float32)
    t.uniform_(-bound, bound, generator=gen)
    return t.to(device).requires_grad_(True)


class PendingLogs:
    """Logs of a deferred ``graphed_train_step``: a pinned host copy of the step's [loss, *metrics]
    and the event that marks it complete.  Slots are reused round-robin (``Conv3DModel.LOG_RING``):
    read a step's logs before that many further steps have been issued."""

    def __init__(self, n):
        self.host = torch.empty((n,), dtype=torch.float32, pin_memory=True)
        self.event = torch.cuda.Event()
        self.names = []

    def result(self):
        self.event.synchronize()
        return dict(zip(self.names, self.host.tolist()))


class Conv3DModel:
    """``create_model(input_shape)`` of convolution.py; ``input_shape = (n_vars, Z, Y, X)``."""

    CONV_SPECS = ((32, 7), (64, 5))                     # (filters, kernel side) (:164,166)

    def __init__(self, input_shape, device=None, seed=0, pool_channels_last=True,
                 broadcast_targets=True):
        self.input_shape = tuple(int(v) for v in input_shape)
        self.device = torch.device(device if device is not None else "cuda")
        self.pool_channels_last = pool_channels_last
        self.broadcast_targets = broadcast_targets
        gen = torch.Generator().manual_seed(seed)
        self.params: dict[str, torch.Tensor] = {}
        shape = (1,) + self.input_shape                 # trace one sample through a branch
        for branch in ("1", "2"):
            cur = shape
            for li, (filters, k) in enumerate(self.CONV_SPECS, start=1):
                cin = cur[1]
                fan_in, fan_out = cin * k ** 3, filters * k ** 3
                name = f"{branch}Conv3D{li}"
                self.params[name + "/kernel"] = _glorot((filters, cin, k, k, k), fan_in, fan_out,
                                                        gen, self.device)
                self.params[name + "/bias"] = torch.zeros((filters,), device=self.device,
                                                          requires_grad=True)
                cur = (1, filters) + tuple(d - k + 1 for d in cur[2:])
                if min(cur[2:]) < 1:
                    raise ValueError(f"input {self.input_shape} too small for kernel {k}")
                cur = self._pooled_shape(cur)
            flat = prod(cur[1:])
        self.flat = 2 * flat
        self.out_dim = prod(self.input_shape)
        if not (ops.dense_supported(self.flat, self.flat) and
                ops.dense_supported(self.flat, self.out_dim)):
            raise ValueError(
                f"input {self.input_shape} gives Dense widths {self.flat} -> {self.flat} -> "
                f"{self.out_dim}; wm_dense_*_f32 takes multiples of 4 floats (16-byte rows) and "
                "this model has no library fallback")
        self.params["dense1/kernel"] = _glorot((self.flat, self.flat), self.flat, self.flat, gen,
                                               self.device)
        self.params["dense1/bias"] = torch.zeros((self.flat,), device=self.device,
                                                 requires_grad=True)
        self.params["output/kernel"] = _glorot((self.out_dim, self.flat), self.flat, self.out_dim,
                                               gen, self.device)
        self.params["output/bias"] = torch.zeros((self.out_dim,), device=self.device,
                                                 requires_grad=True)
        self.optimizer = None
        self._metric_keys: list[str] = []
        self.metrics_names = ["loss"]
        self.stop_training = False

    def _pooled_shape(self, s):
        if self.pool_channels_last:
            return (s[0], s[1] // 2, s[2] // 2, s[3] // 2, s[4])
        return (s[0], s[1], s[2] // 2, s[3] // 2, s[4] // 2)

    # ------------------------------------------------------------------ forward
    def _branch(self, x, branch, last_pool=True):
        for li in (1, 2):
            name = f"{branch}Conv3D{li}"
            x = ops.conv3d(x, self.params[name + "/kernel"], self.params[name + "/bias"])
            if li == 1 or last_pool:
                x = ops.maxpool3d(x, self.pool_channels_last)
        return x.reshape(x.shape[0], -1) if last_pool else x

    # The two conv branches (convolution.py:164-173) are independent until the concatenation and
    # their kernels run 224-workgroup grids on 256 CUs: branch 2 goes to a second stream, forward
    # and -- autograd runs a node's backward on its forward's stream -- backward, so the device can
    # run a kernel of each side by side.  Also under graph capture (the side stream forks from and
    # rejoins the capturing stream through events).
    BRANCH_STREAMS = True
    # Adam for the two Dense kernels launched from their gradient hooks, beside the conv branches'
    # backward pass (DeviceAdam.overlap): measured SLOWER here (0.398 -> 0.424 ms per step: the
    # update is an HBM stream that takes bandwidth from the conv kernels it runs beside), so off.
    OVERLAP_ADAM = False
    # the branches' last pools and the concatenation as one launch that writes the Dense layer's
    # input directly (ops.maxpool3d_cat); needs the channels-last pools the reference runs
    POOL_CAT = True

    def forward(self, x1, x2):
        fused_cat = self.POOL_CAT and self.pool_channels_last and x1.is_cuda
        if self.BRANCH_STREAMS and x1.is_cuda:
            main = torch.cuda.current_stream(x1.device)
            side = self.__dict__.get("_side_stream")
            if side is None or side.device != x1.device:
                side = self._side_stream = torch.cuda.Stream(x1.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                h2 = self._branch(x2, "2", not fused_cat)
            h1 = self._branch(x1, "1", not fused_cat)
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                h2.record_stream(main)
        else:
            h1, h2 = self._branch(x1, "1", not fused_cat), self._branch(x2, "2", not fused_cat)
        h = ops.maxpool3d_cat(h1, h2) if fused_cat else torch.cat((h1, h2), dim=1)
        p = self.params
        # hand-written skinny products; dense1's ReLU backward is applied in the output layer's
        # input-gradient store.  (Widths that are not multiples of 4 floats are refused in
        # __init__: there is no library fallback behind this path.)
        h = ops.dense(h, p["dense1/kernel"], p["dense1/bias"], relu=True, relu_bwd_fused=True)
        return ops.dense(h, p["output/kernel"], p["output/bias"], relu_input=True)

    def _to_dev(self, a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, np.float32))
        t = t.to(self.device, torch.float32)
        want = (t.shape[0],) + self.input_shape
        if tuple(t.shape) != want:
            raise ValueError(f"input has shape {tuple(t.shape)}, model takes {want}")
        return t.contiguous()

    def __call__(self, inputs, training=False):
        x1, x2 = inputs
        if training:
            return self.forward(self._to_dev(x1), self._to_dev(x2))
        with torch.no_grad():
            return self.forward(self._to_dev(x1), self._to_dev(x2))

    def predict(self, x, batch_size=None, verbose=0, **kwargs):
        """``model.predict([data1, data2])`` -> (B, prod(input_shape)) float32 (:233)."""
        return self(x).cpu().numpy()

    # ------------------------------------------------------------------ training
    def compile(self, optimizer="adam", loss=None, metrics=None, weighted_metrics=None, **kwargs):
        name = loss if isinstance(loss, str) or loss is None else getattr(loss, "__name__", "")
        if name not in (None, "mean_squared_error", "mse"):
            raise ValueError(f"only mean_squared_error is implemented (got {name!r})")
        if optimizer not in ("adam", "Adam") and not isinstance(optimizer, torch.optim.Optimizer):
            raise ValueError("optimizer must be 'adam' or a torch optimizer over model.params")
        if isinstance(optimizer, torch.optim.Optimizer):
            self.optimizer = optimizer
        else:
            # one launch for all tensors, Keras' epsilon placement; the two Dense kernels (96 % of
            # the weights, and the first gradients ready) are updated from their gradient hooks on
            # a side stream while the conv branches' backward pass runs
            self.optimizer = ops.DeviceAdam(list(self.params.values()), lr=1e-3, betas=(0.9, 0.999),
                                            eps=1e-7, keras_eps=True)
            if self.OVERLAP_ADAM:
                self.optimizer.overlap([self.params["output/kernel"], self.params["dense1/kernel"]])
        keys = []
        for m in list(metrics or []) + list(weighted_metrics or []):
            m = {"mean_absolute_error": "mae", "mean_squared_logarithmic_error": "msle"}.get(m, m)
            if m not in ("mae", "msle"):
                raise ValueError(f"unsupported metric {m!r}")
            keys.append(m)
        self._metric_keys = keys
        self.metrics_names = ["loss"] + keys
        self.__dict__.pop("_graphs", None)                   # captured steps hold the old optimizer / metrics

    def _targets(self, y):
        t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y, np.float32))
        return t.to(self.device, torch.float32)

    def loss_fn(self, pred, y):
        """keras.losses.mean_squared_error as the reference reaches it: targets arrive as
        (B, n_vars, prod(Z,Y,X)) (:137,214) against predictions (B, P), so the subtraction
        broadcasts to (B, B, P) when n_vars == 1 -- every prediction is scored against every
        target of the batch.  ``broadcast_targets=False`` flattens the targets to (B, P)."""
        if y.dim() == 3 and self.broadcast_targets:
            return ((y - pred) ** 2).mean()
        return ((y.reshape(pred.shape[0], -1) - pred) ** 2).mean()

    def _metric_tensors(self, pred, y):
        """The compiled metrics of a batch as 0-d device tensors, in ``_metric_keys`` order."""
        out = []
        yy = y if (y.dim() == 3 and self.broadcast_targets) else y.reshape(pred.shape[0], -1)
        for key in self._metric_keys:
            if key == "mae":
                out.append((yy - pred).abs().mean())
            else:                                            # msle
                e = torch.log1p(pred.clamp_min(_MSLE_EPS)) - torch.log1p(yy.clamp_min(_MSLE_EPS))
                out.append((e * e).mean())
        return out

    FUSED_MAX_BATCH = 4096      # wm_mse_metrics_f32's limit (float32 sums of B terms per thread)
    LOG_RING = 4                # deferred logs that may be outstanding at once

    def _fused_targets(self, pred, y):
        """The targets as the (B, P) matrix wm_mse_metrics_f32 takes plus its broadcast flag, or
        None for shapes that kernel does not cover (then the tensor expressions above are used)."""
        B, P = pred.shape
        if B > self.FUSED_MAX_BATCH:
            return None
        if y.dim() == 3 and self.broadcast_targets:
            if y.shape[0] != B or y.shape[1] != 1 or y.shape[2] != P:
                return None
            return y.reshape(B, P), True
        if y.numel() != B * P:
            return None
        return y.reshape(B, P), False

    def _loss_stats(self, pred, y, want_grad):
        """[loss, *metrics] as ONE small device tensor and d loss / d pred (or None): one kernel pass
        (ops.mse_metrics) instead of ~20 element-wise launches over the (B, B, P) differences."""
        ft = self._fused_targets(pred, y)
        if ft is not None:
            out3, grad = ops.mse_metrics(pred.detach(), ft[0], ft[1], eps=_MSLE_EPS,
                                         want_grad=want_grad)
            pick = [0] + [{"mae": 1, "msle": 2}[k] for k in self._metric_keys]
            # (no list indexing: that builds its index tensor on the host, which a graph capture
            # does not allow)
            stats = out3[:len(pick)] if pick == [0, 1, 2][:len(pick)] else \
                torch.stack([out3[k] for k in pick])
            return stats, grad
        if want_grad:
            p = pred.detach().requires_grad_(True)
            loss = self.loss_fn(p, y)
            (grad,) = torch.autograd.grad(loss, p)
        else:
            loss, grad = self.loss_fn(pred.detach(), y), None
        with torch.no_grad():
            stats = torch.stack([loss.detach()] + self._metric_tensors(pred.detach(), y))
        return stats, grad

    def _batch_metrics(self, pred, y):
        return {k: float(v) for k, v in zip(self._metric_keys, self._metric_tensors(pred, y))}

    def train_step(self, data):
        """One optimizer step on a batch (x1, x2, y) (convolution.py:46-62)."""
        if self.optimizer is None:
            raise RuntimeError("compile the model before training")
        x1, x2, y = data
        y = self._targets(y)
        self.optimizer.zero_grad(set_to_none=True)
        pred = self.forward(self._to_dev(x1), self._to_dev(x2))
        stats, grad = self._loss_stats(pred, y, True)
        pred.backward(grad)
        self.optimizer.step()
        return dict(zip(["loss"] + list(self._metric_keys), stats.tolist()))

    # ------------------------------------------------------------------ HIP graph replay
    def graphed_train_step(self, data, defer=False):
        """``train_step`` replayed from a HIP graph: the ~60 launches of one step (two conv branches
        forward and backward, pools, the dense GEMMs, Adam) are captured once per batch shape and
        replayed with one host call -- the step is launch-bound otherwise (2.1 ms of which 1.2 ms
        is GPU time at batch 16).  Returns the same logs as ``train_step`` (loss, mae, msle).
        With ``defer`` the logs come back as a ``PendingLogs`` whose ``result()`` waits for the
        step: the caller can stage and launch the next step first (``fit`` reads every step's
        logs one step late), so the host work of a step -- ~90 us of staging during which the GPU
        was idle -- overlaps the previous replay."""
        if self.optimizer is None:
            raise RuntimeError("compile the model before training")
        x1, x2, y = (self._to_dev(data[0]), self._to_dev(data[1]), self._targets(data[2]))
        key = (tuple(x1.shape), tuple(y.shape))
        graphs = self.__dict__.setdefault("_graphs", {})     # one per batch shape (the last batch of
        g = graphs.get(key)                                   # an epoch is usually shorter)
        if g is None:
            g = {"key": key, "x1": x1.clone(), "x2": x2.clone(), "y": y.clone()}

            def body():
                pred = self.forward(g["x1"], g["x2"])
                # loss and metrics leave the graph as ONE small tensor: one host read per step
                stats, grad = self._loss_stats(pred, g["y"], True)
                pred.backward(grad)
                self.optimizer.step()
                return stats, pred.detach()

            # warm up on a side stream (optimizer state, lazily initialised library state), with
            # the parameters and optimizer state restored afterwards
            plist = list(self.params.values())
            saved = [p.detach().clone() for p in plist]
            saved_state = [{k: v.clone() for k, v in self.optimizer.state.get(p, {}).items()
                            if isinstance(v, torch.Tensor)} for p in plist]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.optimizer.zero_grad(set_to_none=True)
                    body()
            torch.cuda.current_stream().wait_stream(side)
            # no gradients exist at capture: backward then CREATES them in the graph's own pool
            # (fixed addresses, rewritten by every replay) instead of zero-filling and adding
            # into twelve existing ones -- two dozen tiny launches fewer per step
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                g["stats"], g["pred"] = body()
            # the graph holds the addresses of the parameters, their gradients and the Adam
            # moments: put the values back in place, never rebind the tensors
            with torch.no_grad():
                for p, q, st in zip(plist, saved, saved_state):
                    p.copy_(q)
                    for k, v in self.optimizer.state[p].items():
                        if isinstance(v, torch.Tensor):
                            v.copy_(st[k]) if k in st else v.zero_()
            g["grads"] = [p.grad for p in plist]
            g["graph"] = graph
            graphs[key] = g
        for p, gr in zip(self.params.values(), g["grads"]):
            p.grad = gr                                  # an eager step in between may rebind them
        ops.copy_multi([g["x1"], g["x2"], g["y"]],                        # one launch, not three
                       [x1.contiguous(), x2.contiguous(), y.contiguous()])
        g["graph"].replay()
        names = ["loss"] + list(self._metric_keys)
        if not defer:
            return dict(zip(names, g["stats"].tolist()))
        # the next replay overwrites g["stats"]: copy it out in stream order, into pinned memory
        ring = self.__dict__.setdefault("_log_ring", [])
        if not ring:
            ring.extend(PendingLogs(len(names)) for _ in range(self.LOG_RING))
        self._log_next = (self.__dict__.get("_log_next", -1) + 1) % self.LOG_RING
        slot = ring[self._log_next]
        if slot.host.numel() != len(names):
            slot.host = torch.empty((len(names),), dtype=torch.float32, pin_memory=True)
        slot.names = names
        slot.host.copy_(g["stats"], non_blocking=True)
        slot.event.record()
        return slot

    def test_step(self, data):
        x1, x2, y = data
        y = self._targets(y)
        with torch.no_grad():
            pred = self.forward(self._to_dev(x1), self._to_dev(x2))
            stats, _ = self._loss_stats(pred, y, False)
            logs = dict(zip(["loss"] + list(self._metric_keys), stats.tolist()))
        return logs

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None,
            validation_data=None, steps_per_epoch=None, validation_steps=None, use_graph=True,
            **kwargs):
        """Iterative Adam training, ``epochs`` x ``steps_per_epoch`` batches (:220-227);
        ``EarlyStopping(monitor=, patience=)`` callbacks are honoured."""
        source = (x[0], x[1], y) if y is not None else x
        it = None
        if callable(source) or hasattr(source, "__next__"):
            if steps_per_epoch is None:
                raise ValueError("steps_per_epoch is required for a generator source")
            it = source() if callable(source) else source
        hist = History(self)
        stoppers = [c for c in (callbacks or []) if hasattr(c, "patience")]
        best = {id(c): (float("inf"), 0) for c in stoppers}
        for epoch in range(int(epochs)):
            sums, count = {}, 0
            batches = (next(it) for _ in range(int(steps_per_epoch))) if it is not None \
                else _batches(source, steps_per_epoch)
            pending = None

            def add(step_logs):
                for k, v in step_logs.items():
                    sums[k] = sums.get(k, 0.0) + v

            for batch in batches:
                if use_graph:
                    # issue step i, then read the logs of step i-1: the host staging of a step
                    # runs while the previous replay is still on the GPU
                    issued = self.graphed_train_step(batch, defer=True)
                    if pending is not None:
                        add(pending.result())
                    pending = issued
                else:
                    add(self.train_step(batch))
                count += 1
            if pending is not None:
                add(pending.result())
            logs = {k: v / max(count, 1) for k, v in sums.items()}
            if validation_data is not None:
                vs, vc = {}, 0
                for batch in _batches(validation_data, validation_steps):
                    for k, v in self.test_step(batch).items():
                        vs["val_" + k] = vs.get("val_" + k, 0.0) + v
                    vc += 1
                logs.update({k: v / max(vc, 1) for k, v in vs.items()})
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(v)
            hist.epoch.append(epoch)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " +
                      " - ".join(f"{k}: {v:.6g}" for k, v in logs.items()))
            stop = False
            for c in stoppers:
                cur = logs.get(c.monitor)
                if cur is None:
                    continue
                b, wait = best[id(c)]
                b, wait = (cur, 0) if cur < b else (b, wait + 1)
                best[id(c)] = (b, wait)
                stop = stop or wait >= c.patience
            if stop:
                break
        self.history = hist
        return hist

    # ------------------------------------------------------------------ weights
    LAYER_ORDER = ("1Conv3D1", "1Conv3D2", "2Conv3D1", "2Conv3D2", "dense1", "output")

    def get_weights(self):
        """[kernel, bias] per layer in creation order, in Keras layouts: conv kernels
        (kD,kH,kW,Cin,Cout), dense kernels (in, out)."""
        out = []
        for name in self.LAYER_ORDER:
            k = self.params[name + "/kernel"].detach()
            k = k.permute(2, 3, 4, 1, 0) if k.dim() == 5 else k.t()
            out += [k.contiguous().cpu().numpy(), self.params[name + "/bias"].detach().cpu().numpy()]
        return out

    def set_weights(self, weights):
        if len(weights) != 2 * len(self.LAYER_ORDER):
            raise ValueError(f"expected {2 * len(self.LAYER_ORDER)} arrays")
        with torch.no_grad():
            for i, name in enumerate(self.LAYER_ORDER):
                k = torch.from_numpy(np.asarray(weights[2 * i], np.float32))
                k = k.permute(4, 3, 0, 1, 2) if k.dim() == 5 else k.t()
                self.params[name + "/kernel"].copy_(k.contiguous())
                self.params[name + "/bias"].copy_(torch.from_numpy(np.asarray(weights[2 * i + 1],
                                                                              np.float32)))

    def count_params(self):
        return sum(p.numel() for p in self.params.values())

    def summary(self, print_fn=print):
        for name in self.LAYER_ORDER:
            print_fn(f"{name:10s} kernel {tuple(self.params[name + '/kernel'].shape)}")
        print_fn(f"total parameters: {self.count_params()}")
