"""Drop-in mirror of the reference's ``model.py``: the XOR toy network.

``createUncompiledModel()``, ``createCompiledModel()``, ``createDataset(datasetSize)`` as in
/root/reference/model.py:33-72.  The reference also builds, trains and evaluates the model AT
IMPORT (:32,74-94); that script part lives in ``main()`` here, so importing has no side effects.

The network (two scalar inputs -> concatenate -> three Dense(2, tanh) -> Dense(1, tanh), :34-42) is
not on the accelerated hot path (SURVEY 8a lists no row for it): it is 21 parameters trained by
batch-2 SGD, latency-bound on any device, and runs as plain host torch code.
"""
from __future__ import annotations

import random

import numpy as np
import torch
from torch import nn

from ._keras_like import History


class _ToyModel:
    """The slice of the keras.Model surface model.py uses: compile / fit / evaluate / predict."""

    def __init__(self):
        gen = torch.Generator().manual_seed(0)
        self.net = nn.Sequential(nn.Linear(2, 2), nn.Tanh(), nn.Linear(2, 2), nn.Tanh(),
                                 nn.Linear(2, 2), nn.Tanh(), nn.Linear(1 * 2, 1), nn.Tanh())
        for m in self.net:
            if isinstance(m, nn.Linear):              # keras Dense default: glorot-uniform, zero bias
                bound = (6.0 / (m.in_features + m.out_features)) ** 0.5
                with torch.no_grad():
                    m.weight.uniform_(-bound, bound, generator=gen)
                    m.bias.zero_()
        self.layer_names = ("layer1", "layer2", "layer3", "output")
        self.lr = 0.01                                # keras.optimizers.SGD() default
        self.compiled = False
        self.metrics_names = ["loss"]

    def compile(self, optimizer=None, loss=None, metrics=None, **kwargs):
        lr = getattr(optimizer, "learning_rate", None)
        if isinstance(lr, (int, float)):
            self.lr = float(lr)
        self.metrics_names = ["loss", "mean_squared_error", "accuracy"]
        self.compiled = True

    @staticmethod
    def _xy(x, y=None):
        if isinstance(x, (list, tuple)):
            x = np.concatenate([np.asarray(v, np.float32).reshape(len(v), -1) for v in x], axis=1)
        xt = torch.as_tensor(np.asarray(x, np.float32)).reshape(-1, 2)
        yt = None if y is None else torch.as_tensor(np.asarray(y, np.float32)).reshape(-1, 1)
        return xt, yt

    def _score(self, xt, yt):
        with torch.no_grad():
            pred = self.net(xt)
            mse = float(((pred - yt) ** 2).mean())
            acc = float((pred == yt).float().mean())   # keras.metrics.Accuracy: exact equality
        return [mse, mse, acc]

    def fit(self, x, y, batch_size=32, epochs=1, validation_split=0.0, verbose=1, **kwargs):
        xt, yt = self._xy(x, y)
        n_val = int(len(xt) * validation_split)       # keras holds out the LAST fraction
        n_tr = len(xt) - n_val
        opt = torch.optim.SGD(self.net.parameters(), lr=self.lr)
        hist = History(self)
        perm_gen = torch.Generator().manual_seed(1)
        for epoch in range(epochs):
            order = torch.randperm(n_tr, generator=perm_gen)
            total, count = 0.0, 0
            for s in range(0, n_tr, batch_size):
                idx = order[s:s + batch_size]
                opt.zero_grad()
                loss = ((self.net(xt[idx]) - yt[idx]) ** 2).mean()
                loss.backward()
                opt.step()
                total += float(loss.detach()) * len(idx)
                count += len(idx)
            logs = {"loss": total / max(count, 1)}
            tr = self._score(xt[:n_tr], yt[:n_tr])
            logs["mean_squared_error"], logs["accuracy"] = tr[1], tr[2]
            if n_val:
                va = self._score(xt[n_tr:], yt[n_tr:])
                logs.update({"val_loss": va[0], "val_mean_squared_error": va[1],
                             "val_accuracy": va[2]})
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(v)
            hist.epoch.append(epoch)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " +
                      " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
        return hist

    def evaluate(self, x, y, batch_size=32, verbose=0, **kwargs):
        return self._score(*self._xy(x, y))

    def predict(self, x, **kwargs):
        xt, _ = self._xy(x)
        with torch.no_grad():
            return self.net(xt).numpy()

    def __call__(self, x, training=False):
        return self.predict(x)

    def summary(self, print_fn=print):
        print_fn(f"toy tanh MLP 2-2-2-2-1, layers {self.layer_names}, "
                 f"{sum(p.numel() for p in self.net.parameters())} parameters")


def createUncompiledModel():
    return _ToyModel()                                                      # :33-43


def createCompiledModel():
    model = createUncompiledModel()
    model.compile(optimizer=None, loss='mean_squared_error',
                  metrics=['mean_squared_error', 'accuracy'])
    return model                                                            # :45-53


def createDataset(datasetSize):
    """XOR truth-table samples: a leading (0,0)->0 row plus datasetSize random rows (:55-72)."""
    table = {0: ((0, 0), 0), 1: ((0, 1), 1), 2: ((1, 0), 1), 3: ((1, 1), 0)}
    picks = [random.randint(0, 3) for _ in range(datasetSize)]
    x = np.array([[0, 0]] + [table[v][0] for v in picks]).reshape(-1, 2)
    y = np.array([[0]] + [[table[v][1]] for v in picks]).reshape(-1, 1)
    return x, y


def main():
    """What the reference runs at import (:74-94)."""
    print('Testing the functional version now')
    model = createCompiledModel()
    datasetSize, batchSize, epochs = 20000, 2, 4
    x_train, y_train = createDataset(datasetSize)
    x_test, y_test = createDataset(int(datasetSize / 10))
    history = model.fit(x_train, y_train, batch_size=batchSize, epochs=epochs,
                        validation_split=0.2)
    print(history.history)
    print(model.evaluate(x_test, y_test, batch_size=batchSize))
    print(model.predict(np.array([[0, 0], [1, 0], [0, 1], [1, 1]])))


if __name__ == '__main__':
    main()
