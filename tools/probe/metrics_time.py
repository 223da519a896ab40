"""Time the drop-ins' per-epoch metrics pass: the fused kernel vs predict + framework reductions."""
import sys, time
import torch
from weather_model_amd import ops

def old(X, Y, beta):
    pred = ops.linear_predict(X, beta).double()
    yv = Y.double()
    d = pred - yv
    a = (d * d).sum(); b = d.abs().sum()
    e = torch.log1p(pred.clamp_min(1e-7)) - torch.log1p(yv.clamp_min(1e-7))
    return a, b, (e * e).sum()

for spec in sys.argv[1:] or ["2:1:40000000", "16:8:40000000", "12:6:40000000", "32:1:20000000", "432:216:400000"]:
    F, P, n = (int(v) for v in spec.split(":"))
    X = ops.synth_uniform((n, F), seed=1)
    # a fitted model's rows: Y = [X | 1] beta + small noise (targets unrelated to the predictions
    # would put every output on the rare far form of the msle term)
    beta = (torch.rand((F + 1, P), device="cuda", dtype=torch.float64) - 0.5) / F ** 0.5
    beta[F] += 1.0
    Y = ops.synth_uniform((n, P), seed=2, lo=-0.02, hi=0.02)
    step = max(1, 200_000_000 // (F + P))
    for a in range(0, n, step):
        Y[a:a + step] += (X[a:a + step].double() @ beta[:F] + beta[F]).float()
    for name, fn in (("fused", lambda: ops.linear_metrics(X, Y, beta)),
                     ("nomsle", lambda: ops.linear_metrics(X, Y, beta, want_msle=False)), ("old", lambda: old(X, Y, beta)),
                     ("gram", lambda: ops.gram(X, Y))):
        if name == "old" and n * P > 4e8:
            continue
        fn(); torch.cuda.synchronize()
        reps = 5
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        gb = n * (F + P) * 4 / 1e9
        print(f"F={F} P={P} n={n} {name:6s} {ms:8.3f} ms  {n / ms / 1e6:7.2f} G rows/s  {gb / ms * 1e3:7.0f} GB/s", flush=True)
