"""The model object the regression drop-ins hand out in place of a ``keras.Model``.

The reference builds ``[x1, x2] -> Flatten -> Concatenate -> Dense(linear) x k -> Dense`` and fits
it with Adam on MSE (multipleRegression.py:273-298,341-350; singleRegression.py:74-96,500-509).
Every layer is linear with a bias, so the stack spans exactly the affine maps R^{2p} -> R^p and the
minimiser of its loss is ordinary least squares with an intercept.  ``LinearStackModel.fit``
therefore consumes the same ``(x1, x2, y)`` batches, accumulates the Gram of ``[x1|x2|1|y]`` on the
GPU (``wm_gram_f32``), solves the normal equations once (``wm_normal_solve_f64``) and reports the
loss/metrics of that exact minimiser; ``predict`` is ``wm_linear_predict_f32``.

What a caller of the reference sees unchanged: ``compile(...)``, ``fit(dataset, epochs=,
steps_per_epoch=, validation_data=, validation_steps=, batch_size=, callbacks=, verbose=)`` ->
object with ``.history`` (dict of lists) and ``.epoch`` (list), ``predict([x1, x2])`` -> float32
ndarray ``(B, prod(input_shape))``, ``evaluate``, ``save``/``export``, ``get_weights`` (the Keras
variable order: kernel, bias per Dense layer).  What differs: the SOLVER.  Epoch e consumes the next
``steps_per_epoch`` batches of the (never-ending, region-walking) generator exactly as Keras would,
adds their Gram to the running Gram of all epochs so far and re-solves, so after E epochs the map
is the least-squares fit over the same ``E * steps_per_epoch`` batches the reference's Adam loop
would have seen; ``history['loss'][e]`` is the MSE of that epoch's solution on that epoch's rows,
and ``EarlyStopping(monitor, patience)`` acts on it as in Keras.  ``History.params`` names the
solver, the ridge used and the parity status so that nobody mistakes the numbers for an Adam
trajectory.  There is no CPU path: without libwm.so and a GPU every compute call raises.

Rows never have to fit in memory twice: a generator that can rebuild its batches on the GPU from a
resident volume (``device_rows``: ``wm_gather_cubes_f32``) is streamed chunk by chunk; host
generators are staged through a pinned buffer.  Either way the device chunks are kept for the
metrics pass up to ``max_resident_rows`` (and 8 GiB for the gathered ones); beyond that a device
generator is rewound and gathered again, a host generator's metrics fall back to the loss from the
Gram.

Parity note: tensorflow/keras are absent from the build container, so nothing here is pinned
against the reference's own training run ("parity unpinned"); the fit is pinned against float64
``numpy.linalg.lstsq`` on the same rows (tests/test_gpu_regression_dropin.py).
"""
from __future__ import annotations

import json
import os
from math import prod

import numpy as np
import torch

from weather_model_amd import dp, ops

_MSLE_EPS = 1e-7          # keras.backend.epsilon(), the clip of mean_squared_logarithmic_error


class History:
    """Stand-in for ``keras.callbacks.History``: ``.history`` and ``.epoch`` are what the reference
    reads (multipleRegression.py:559-560, singleRegression.py:559)."""

    def __init__(self, model=None):
        self.model = model
        self.history: dict[str, list[float]] = {}
        self.epoch: list[int] = []
        self.params: dict = {}

    def __repr__(self):
        return f"History(epoch={self.epoch}, history={self.history})"


class EarlyStopping:
    """``tf.keras.callbacks.EarlyStopping(monitor='loss', patience=2)`` (multipleRegression.py:339,
    singleRegression.py:498): stop once ``monitor`` has not improved by more than ``min_delta`` for
    ``patience`` consecutive epochs (minimised quantity)."""

    def __init__(self, monitor="val_loss", patience=0, min_delta=0.0, **kwargs):
        self.monitor, self.patience, self.min_delta = monitor, int(patience), abs(float(min_delta))
        self.kwargs = kwargs
        self.best, self.wait, self.stopped_epoch = float("inf"), 0, 0

    def reset(self):
        self.best, self.wait, self.stopped_epoch = float("inf"), 0, 0

    def update(self, epoch, logs) -> bool:
        """True when training should stop after this epoch."""
        value = logs.get(self.monitor)
        if value is None or value != value:
            return False
        if value < self.best - self.min_delta:
            self.best, self.wait = value, 0
            return False
        self.wait += 1
        if self.wait >= self.patience:
            self.stopped_epoch = epoch
            return True
        return False


def _metric_names(metrics):
    names = []
    for m in metrics or []:
        n = m if isinstance(m, str) else getattr(m, "__name__", getattr(m, "name", str(m)))
        n = {"mean_absolute_error": "mae", "mean_squared_logarithmic_error": "msle",
             "mse": "mean_squared_error"}.get(n, n)
        if n not in ("mae", "msle", "mean_squared_error"):
            raise ValueError(f"unsupported metric {n!r} (have mae, msle, mean_squared_error)")
        if n not in names:
            names.append(n)
    return names


def _as_rows(x1, x2, y):
    """One reference batch -> (X (B,2p) float32, Y (B,p) float32) host arrays."""
    a = np.asarray(x1, dtype=np.float32)
    b = np.asarray(x2, dtype=np.float32)
    t = np.asarray(y, dtype=np.float32)
    if a.shape != b.shape:
        raise ValueError(f"x1 {a.shape} and x2 {b.shape} differ")
    B = a.shape[0]
    if t.shape[0] != B:
        raise ValueError(f"y has {t.shape[0]} rows, x has {B}")
    return np.concatenate((a.reshape(B, -1), b.reshape(B, -1)), axis=1), t.reshape(B, -1)


def _batches(source, steps):
    """Iterate the (x1, x2, y) batches of a reference-style data source: a callable returning a
    generator (WeatherDataGenerator / WeatherGenerator, which the reference wraps with
    tf.data.Dataset.from_generator), a plain iterator, a Sequence (``len`` + ``[i]``), or one
    ``(x1, x2, y)`` tuple of arrays."""
    if isinstance(source, (tuple, list)) and len(source) == 3 and not callable(source[0]):
        yield source
        return
    if hasattr(source, "__len__") and hasattr(source, "__getitem__") and not callable(source):
        n = len(source) if steps is None else min(steps, len(source))
        for i in range(n):
            yield source[i]
        return
    it = source() if callable(source) else iter(source)
    if steps is None:
        raise ValueError("steps_per_epoch / validation_steps is required for a generator source "
                         "(the reference's generators never end)")
    for _ in range(int(steps)):
        try:
            yield next(it)
        except StopIteration:
            return


class _RowStore:
    """Rows staged through one pinned host buffer into resident device chunks."""

    def __init__(self, F, P, device, chunk_rows, on_chunk=None, max_resident_rows=None):
        self.F, self.P, self.device = F, P, device
        self.chunk_rows = int(chunk_rows)
        self.host = torch.empty((self.chunk_rows, F + P), dtype=torch.float32,
                                pin_memory=torch.cuda.is_available())
        self.fill = 0
        self.chunks: list[tuple[torch.Tensor, torch.Tensor]] = []
        self.rows = 0
        self.on_chunk = on_chunk                      # called with (X, Y) as each chunk lands
        self.max_resident_rows = max_resident_rows    # beyond it chunks are dropped after on_chunk
        self.dropped = False

    def add(self, X, Y):
        if X.shape[1] != self.F or Y.shape[1] != self.P:
            raise ValueError(f"batch is ({X.shape[1]} -> {Y.shape[1]}), model is "
                             f"({self.F} -> {self.P})")
        pos, B = 0, X.shape[0]
        while pos < B:
            m = min(B - pos, self.chunk_rows - self.fill)
            dst = self.host[self.fill:self.fill + m]
            dst[:, :self.F] = torch.from_numpy(X[pos:pos + m])
            dst[:, self.F:] = torch.from_numpy(Y[pos:pos + m])
            self.fill += m
            pos += m
            if self.fill == self.chunk_rows:
                self.flush()

    def flush(self):
        if self.fill == 0:
            return
        dev = self.host[:self.fill].to(self.device, non_blocking=True)
        X = dev[:, :self.F].contiguous()
        Y = dev[:, self.F:].contiguous()
        torch.cuda.current_stream(self.device).synchronize()     # host buffer is reused next
        if self.on_chunk is not None:
            self.on_chunk(X, Y)
        self.rows += self.fill
        if self.max_resident_rows is not None and self.rows > self.max_resident_rows:
            self.dropped = True
            self.chunks.clear()
        elif not self.dropped:
            self.chunks.append((X, Y))
        self.fill = 0


class LinearStackModel:
    """``create_model`` / ``createModel`` of the regression scripts, as one affine map."""

    def __init__(self, input_shape, hidden_layers=("layer1", "layer2"), output_name="output",
                 device=None):
        self.input_shape = tuple(int(v) for v in input_shape)
        self.p = prod(self.input_shape)
        self.F, self.P = 2 * self.p, self.p
        self.layer_names = tuple(hidden_layers) + (output_name,)
        self.device = torch.device(device if device is not None else "cuda")
        self.beta = None            # (F+1, P) float64 device tensor, intercept last
        self.ridge = 0.0
        self.compiled = False
        self.metrics_names: list[str] = ["loss"]
        self._metric_keys: list[str] = []
        self.optimizer = None
        self.loss = None
        self.stop_training = False
        self.rows_seen = 0

    # ------------------------------------------------------------------ keras surface
    def compile(self, optimizer="adam", loss=None, metrics=None, weighted_metrics=None, **kwargs):
        """Records the configuration.  The optimizer is irrelevant to the result (the fit goes to
        the minimiser Adam/SGD converge to); only mean-squared-error loss is meaningful."""
        name = loss if isinstance(loss, str) else getattr(loss, "__name__", None)
        if loss is not None and name not in ("mean_squared_error", "mse", "MSE", None):
            raise ValueError(f"only the mean-squared-error loss has a closed form (got {name!r})")
        self.optimizer, self.loss = optimizer, loss
        self._metric_keys = _metric_names(list(metrics or []) + list(weighted_metrics or []))
        self.metrics_names = ["loss"] + self._metric_keys
        self.compiled = True
        return None

    def _need_beta(self):
        if self.beta is None:
            # an unfitted keras model predicts from its random initial weights; ours from zeros
            self.beta = torch.zeros((self.F + 1, self.P), device=self.device, dtype=torch.float64)
        return self.beta

    def _device_rows(self, x1, x2):
        if isinstance(x1, torch.Tensor):
            a = x1.to(self.device, torch.float32)
            b = x2.to(self.device, torch.float32)
            return torch.cat((a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)), 1).contiguous()
        a = np.asarray(x1, dtype=np.float32)
        b = np.asarray(x2, dtype=np.float32)
        X = np.concatenate((a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)), axis=1)
        return torch.from_numpy(np.ascontiguousarray(X)).to(self.device)

    def __call__(self, inputs, training=False):
        x1, x2 = inputs
        X = self._device_rows(x1, x2)
        if X.shape[1] != self.F:
            raise ValueError(f"inputs flatten to {X.shape[1]} columns, model takes {self.F}")
        return ops.linear_predict(X, self._need_beta())

    def predict(self, x, batch_size=None, verbose=0, steps=None, **kwargs):
        """``model.predict([data1, data2])`` -> float32 ndarray (B, prod(input_shape))
        (multipleRegression.py:451, singleRegression.py:579)."""
        return self(x).cpu().numpy()

    # ------------------------------------------------------------------ fit
    def _metrics(self, chunks, prefix=""):
        """loss + compiled metrics of the current beta over resident row chunks."""
        sums = self._new_sums()
        for X, Y in chunks:
            self._sums_of(X, Y, sums)
        return self._finish_sums(sums, prefix)

    def _sums_of(self, X, Y, sums):
        """Add the residual sums of the current beta over one resident chunk: one fused pass
        (linmetrics.hip) -- the predictions are never written."""
        ops.linear_metrics(X, Y, self.beta, sums, want_msle="msle" in self._metric_keys,
                           msle_eps=_MSLE_EPS)

    def _sums_of_walk(self, vol, d_mid, d_end, rows, walk, sums):
        """``_sums_of`` over rows that are never built (linear_metrics_walk_kernel)."""
        ops.linear_metrics_walk(vol, d_mid, d_end, rows, walk, self.beta, sums,
                                want_msle="msle" in self._metric_keys, msle_eps=_MSLE_EPS)

    _SUM_SLOT = {"loss": 0, "mean_squared_error": 0, "mae": 1, "msle": 2}

    def _finish_sums(self, sums, prefix=""):
        keys = ["loss"] + self._metric_keys
        both = sums[[self._SUM_SLOT[k] for k in keys] + [3]]
        if dp.world() > 1:
            dp.allreduce_sum_(both)
        vals = (both[:-1] / both[-1].clamp_min(1.0)).cpu().tolist()
        return {prefix + k: float(v) for k, v in zip(keys, vals)}

    def _new_sums(self):
        """{sum d^2, sum |d|, sum (log1p diff)^2, count} on the device, float64."""
        return torch.zeros((4,), device=self.device, dtype=torch.float64)

    def _loss_from_gram(self, acc):
        """MSE of the current beta over the rows summarised by ``acc`` (the Gram of [X|1|Y]):
        (tr Y'Y - 2 tr beta'Z'Y + tr beta'Z'Z beta) / (n P).  Used only when the rows themselves are
        gone; it inherits the float32 rounding of the Gram products (~1e-7 of tr Y'Y)."""
        F1 = self.F + 1
        A, B, C = acc[:F1, :F1], acc[:F1, F1:], acc[F1:, F1:]
        sse = torch.trace(C) - 2.0 * (self.beta * B).sum() + (self.beta * (A @ self.beta)).sum()
        n = acc[self.F, self.F]
        return float((sse.clamp_min(0.0) / (n * self.P).clamp_min(1.0)).item())

    def _walk_source(self, source):
        """True when ``source`` can hand its batches over as a walk of the resident volume
        (``device_walk``) that the fused kernels take for this model's shape."""
        if os.environ.get("WM_FIT_WALK", "1") == "0" or not hasattr(source, "device_walk"):
            return False
        return bool(source.device_walk_ready(self.device, self.F, self.P))

    def _stream(self, source, steps, chunk_rows, on_chunk, keep_rows_up_to, on_walk=None):
        """Feed ``steps`` batches of ``source`` to ``on_chunk(X, Y)`` in device chunks.  Returns a
        callable ``replay(fn, walk_fn=None)`` that feeds the same chunks again (for the metrics
        pass) or None if they are gone.
        ``on_walk(vol, d_mid, d_end, rows, walk)``: for a generator over a resident volume whose
        rows the kernels can read themselves (``device_walk``: the reference's CUBE_SIZE 1,
        multipleRegression.py:35) no rows are built at all -- ``on_walk`` gets the walk instead, and
        the replay calls ``walk_fn`` the same way."""
        device_gen = self.device.type == "cuda" and hasattr(source, "device_rows") and \
            getattr(source, "device_ready", lambda: False)()
        if device_gen and steps is None:
            raise ValueError("steps_per_epoch / validation_steps is required for a generator")
        if device_gen and on_walk is not None and self._walk_source(source):
            bs = max(1, int(source.rows_per_batch()))
            per = max(1, (1 << 30) // bs)             # the kernels count rows in 31 bits
            state = source.get_state()

            def run_walk(fn):
                left = int(steps)
                while left > 0:
                    k = min(per, left)
                    fn(*source.device_walk(k, self.device))
                    left -= k
            run_walk(on_walk)
            end_state = source.get_state()

            def replay(fn, walk_fn=None):
                if walk_fn is None:
                    raise RuntimeError("the rows of this pass were never built: replay needs walk_fn")
                source.set_state(state)
                run_walk(walk_fn)
                source.set_state(end_state)
            self.rows_from = "volume (wm_gram_walk_f32 / wm_linear_metrics_walk_f32)"
            return replay
        if device_gen:
            self.rows_from = "gathered on the device (wm_gather_cubes_f32)"
            if chunk_rows is None:
                # rows built on the device: chunks of ~1 GiB, so that the two launches per chunk
                # (gather, Gram or metrics: ~0.2 ms of GPU time) outlast the host work between them
                chunk_rows = max(1 << 16, (1 << 30) // (4 * (self.F + self.P)))
            per = max(1, int(chunk_rows) // max(1, int(source.rows_per_batch())))
            state = source.get_state()
            # the gathered chunks stay resident for the metrics pass while they fit the row budget
            # and 8 GiB: at scale the gather is 2-3x the Gram or metrics pass it feeds, and until
            # round 3's last session an epoch gathered every chunk twice
            kept: list = []
            budget = {"rows": int(keep_rows_up_to or 0), "bytes": 8 << 30, "ok": bool(keep_rows_up_to)}

            def run(fn, collect=False):
                left = int(steps)
                while left > 0:
                    k = min(per, left)
                    X, Y = source.device_rows(k, self.device)
                    fn(X, Y)
                    if collect and budget["ok"]:
                        budget["rows"] -= X.shape[0]
                        budget["bytes"] -= 4 * X.shape[0] * (self.F + self.P)
                        if budget["rows"] < 0 or budget["bytes"] < 0:
                            budget["ok"] = False
                            kept.clear()
                        else:
                            kept.append((X, Y))
                    left -= k
            run(on_chunk, collect=True)
            end_state = source.get_state()

            def replay(fn, walk_fn=None):
                if budget["ok"]:
                    for X, Y in kept:
                        fn(X, Y)
                    return
                source.set_state(state)
                run(fn)
                source.set_state(end_state)
            return replay
        store = _RowStore(self.F, self.P, self.device, chunk_rows or 1 << 18, on_chunk,
                          keep_rows_up_to)
        for batch in _batches(source, steps):
            store.add(*_as_rows(*batch))
        store.flush()
        if store.dropped:
            return None
        chunks = store.chunks

        def replay(fn, walk_fn=None):
            for X, Y in chunks:
                fn(X, Y)
        return replay

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None,
            validation_split=0.0, validation_data=None, shuffle=True, steps_per_epoch=None,
            validation_steps=None, chunk_rows=None, ridge=None,
            max_resident_rows=1 << 26, **kwargs):
        """Least-squares fit, one Gram pass + one solve per epoch (see the module docstring).

        ``x`` is any source ``_batches`` understands, or a generator object with ``device_rows``;
        ``(x, y)`` arrays are also accepted with ``x = [x1, x2]``.  Under an initialised
        ``torch.distributed`` group every rank feeds its own batches and the partial Grams are
        summed by one all-reduce per epoch."""
        source = (x[0], x[1], y) if y is not None else x
        fixed_rows = isinstance(source, (tuple, list)) or \
            (hasattr(source, "__len__") and hasattr(source, "__getitem__") and not callable(source))
        epochs = max(1, int(epochs))
        D = self.F + 1 + self.P
        total = torch.zeros((D, D), device=self.device, dtype=torch.float64)
        acc_e = torch.zeros_like(total)
        hist = History(self)
        stoppers = [c for c in (callbacks or []) if isinstance(c, EarlyStopping)]
        for c in stoppers:
            c.reset()
        replay = None
        self.rows_seen = 0
        self.rows_from = "host batches"          # _stream overwrites it for a device generator
        for epoch in range(epochs):
            if not (fixed_rows and epoch > 0):
                # the same arrays every epoch add nothing to a least-squares fit: their Gram is
                # taken once; a generator yields NEW batches every epoch
                acc_e.zero_()
                replay = self._stream(source, steps_per_epoch, chunk_rows,
                                      lambda X, Y: ops.gram(X, Y, acc_e), max_resident_rows,
                                      on_walk=lambda *a: ops.gram_walk(*a, acc=acc_e))
                dp.reduce_gram_(acc_e)
                n_epoch = float(acc_e[self.F, self.F].item())     # sum of 1*1 = global row count
                if n_epoch < 1:
                    raise ValueError("fit received no rows")
                total += acc_e
                self.rows_seen += int(n_epoch)
                self.beta, self.ridge = self._solve(total, ridge)
            if replay is not None:
                sums = self._new_sums()
                replay(lambda X, Y: self._sums_of(X, Y, sums),
                       lambda *a: self._sums_of_walk(*a, sums))
                logs = self._finish_sums(sums)
            else:
                logs = {"loss": self._loss_from_gram(acc_e)}
                logs.update({k: float("nan") for k in self._metric_keys})
                if "mean_squared_error" in logs:
                    logs["mean_squared_error"] = logs["loss"]
            if validation_data is not None:
                sums = self._new_sums()
                self._stream(validation_data, validation_steps, chunk_rows,
                             lambda X, Y: self._sums_of(X, Y, sums), 0,
                             on_walk=lambda *a: self._sums_of_walk(*a, sums))
                logs.update(self._finish_sums(sums, prefix="val_"))
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(v)
            hist.epoch.append(epoch)
            if verbose and dp.rank() == 0:
                shown = " - ".join(f"{k}: {v:.6g}" for k, v in logs.items())
                print(f"Epoch {epoch + 1}/{epochs} (closed-form least squares over "
                      f"{self.rows_seen} rows so far; parity unpinned)\n"
                      f"{steps_per_epoch if steps_per_epoch is not None else '?'} steps - {shown}")
            if any(c.update(epoch, logs) for c in stoppers):
                break
        hist.params = {"epochs": epochs, "steps": steps_per_epoch, "verbose": verbose,
                       "solver": "closed-form ordinary least squares on the running Gram of all "
                                 "epochs (not an Adam trajectory)",
                       "ridge": self.ridge, "rows": self.rows_seen,
                       "metrics_from": "rows" if replay is not None else "gram (rows not kept)",
                       "rows_from": getattr(self, "rows_from", "host batches"),
                       "parity": "unpinned: keras/tensorflow absent, no reference fixture; pinned "
                                 "against float64 numpy.linalg.lstsq on the same rows"}
        self.history = hist
        return hist

    def _solve(self, acc, ridge):
        """Normal-equation solve; if the centred X'X is not numerically positive definite (the
        6x6x6 cubes of singleRegression are strongly collinear) retry with a small ridge relative
        to the mean diagonal, growing tenfold, and report the value used."""
        F, P = self.F, self.P
        if ridge is not None:
            return ops.normal_solve(acc, F, P, True, float(ridge)), float(ridge)
        try:
            return ops.normal_solve(acc, F, P, True, 0.0), 0.0
        except RuntimeError:
            pass
        n = acc[F, F]
        diag = torch.diagonal(acc)[:F] - acc[F, :F] ** 2 / n      # centred sums of squares
        scale = float(diag.clamp_min(0).mean().item())
        scale = scale if scale > 0 else 1.0
        r = 1e-12 * scale
        while r <= 1e-2 * scale:
            try:
                return ops.normal_solve(acc, F, P, True, r), r
            except RuntimeError:
                r *= 10.0
        raise RuntimeError("normal equations stayed singular up to ridge = 1e-2 * mean variance")

    def evaluate(self, x=None, y=None, batch_size=None, verbose=0, steps=None, **kwargs):
        """[loss, *metrics] of the current map over a data source (model.py:88 style)."""
        self._need_beta()
        source = (x[0], x[1], y) if y is not None else x
        if self.device.type == "cuda" and hasattr(source, "device_rows") and \
                getattr(source, "device_ready", lambda: False)():
            # a generator over a resident volume: the same device passes fit's validation uses
            sums = self._new_sums()
            self._stream(source, steps, None, lambda X, Y: self._sums_of(X, Y, sums), 0,
                         on_walk=lambda *a: self._sums_of_walk(*a, sums))
            logs = self._finish_sums(sums)
            out = [logs[k] for k in ["loss"] + self._metric_keys]
            return out if len(out) > 1 else out[0]
        store = _RowStore(self.F, self.P, self.device, 1 << 16)
        for batch in _batches(source, steps):
            store.add(*_as_rows(*batch))
        store.flush()
        logs = self._metrics(store.chunks)
        out = [logs[k] for k in ["loss"] + self._metric_keys]
        return out if len(out) > 1 else out[0]

    # ------------------------------------------------------------------ weights / persistence
    @property
    def coefficients(self):
        """(kernel (2p, p), bias (p,)) float64 ndarrays of the fitted affine map."""
        b = self._need_beta().cpu().numpy()
        return b[:self.F].copy(), b[self.F].copy()

    def get_weights(self):
        """Keras variable order [kernel, bias] per Dense layer; the hidden square layers carry the
        identity so the stack composes to the fitted map (one of its infinitely many equivalent
        factorisations)."""
        K, c = self.coefficients
        out = []
        for _ in self.layer_names[:-1]:
            out += [np.eye(self.F, dtype=np.float32), np.zeros((self.F,), np.float32)]
        return out + [K.astype(np.float32), c.astype(np.float32)]

    def set_weights(self, weights):
        """Accepts the Keras list and stores the composite affine map it spans."""
        if len(weights) != 2 * len(self.layer_names):
            raise ValueError(f"expected {2 * len(self.layer_names)} arrays")
        A = np.eye(self.F, dtype=np.float64)
        c = np.zeros((self.F,), np.float64)
        for i in range(len(self.layer_names)):
            W = np.asarray(weights[2 * i], np.float64)
            b = np.asarray(weights[2 * i + 1], np.float64)
            A = A @ W
            c = c @ W + b
        if A.shape != (self.F, self.P):
            raise ValueError(f"weights compose to {A.shape}, model is {(self.F, self.P)}")
        beta = np.concatenate((A, c[None, :]), axis=0)
        self.beta = torch.from_numpy(beta).to(self.device)

    @property
    def trainable_variables(self):
        return self.get_weights()

    def count_params(self):
        return sum(int(w.size) for w in self.get_weights())

    def summary(self, print_fn=print):
        print_fn(f"LinearStackModel: inputs 2 x {self.input_shape} -> Dense layers "
                 f"{self.layer_names} -> {self.P}; {self.count_params()} parameters; "
                 f"fitted={'yes' if self.beta is not None else 'no'}")

    def save(self, path, **kwargs):
        """``model.save(...)`` (singleRegression.py:616): the affine map as .npz + a JSON config
        (not a .keras archive -- there is no keras here)."""
        K, c = self.coefficients
        base = str(path)
        if os.path.isdir(base):
            base = os.path.join(base, "linear_stack_model")
        np.savez(base if base.endswith(".npz") else base + ".npz", kernel=K, bias=c,
                 config=json.dumps({"input_shape": self.input_shape,
                                    "layers": self.layer_names, "ridge": self.ridge}))

    export = save           # ``mod.export(...)`` (multipleRegression.py:585)

    @classmethod
    def load(cls, path, device=None):
        z = np.load(path, allow_pickle=False)
        cfg = json.loads(str(z["config"]))
        m = cls(tuple(cfg["input_shape"]), tuple(cfg["layers"][:-1]), cfg["layers"][-1], device)
        beta = np.concatenate((z["kernel"], z["bias"][None, :]), axis=0)
        m.beta = torch.from_numpy(beta).to(m.device)
        m.ridge = cfg.get("ridge", 0.0)
        return m
