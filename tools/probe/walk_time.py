"""Times the fit passes of the reference's CUBE_SIZE-1 regression over a resident point-major volume:
gather + Gram / gather + metrics against the passes that read the volume directly."""
import sys
import time

import torch

from weather_model_amd import ops


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    for nv in (1, 2, 3, 4):
        files, Z, Y, X = 4200, 16, 32, 32
        stored = torch.rand((Z, Y, X, files, nv), device="cuda")
        vol = stored.permute(3, 4, 0, 1, 2)
        bs = 64
        length = (files - 6) // bs
        walk = {"batch": bs, "length": length, "index": 0, "region": (0, 0, 0), "dims": (Z, Y, X)}
        rows = bs * length * Z * Y * X
        beta = torch.rand((2 * nv + 1, nv), dtype=torch.float64, device="cuda") / (2 * nv)
        acc = torch.zeros((3 * nv + 1, 3 * nv + 1), dtype=torch.float64, device="cuda")
        sums = torch.zeros((4,), dtype=torch.float64, device="cuda")
        tg = timed(lambda: ops.gather_cubes(vol, (1, 1, 1), 3, 6, rows, walk=walk))
        Xr, Yr = ops.gather_cubes(vol, (1, 1, 1), 3, 6, rows, walk=walk)
        tgram = timed(lambda: ops.gram(Xr, Yr, acc))
        tmet = timed(lambda: ops.linear_metrics(Xr, Yr, beta, sums))
        del Xr, Yr
        twg = timed(lambda: ops.gram_walk(vol, 3, 6, rows, walk, acc))
        twm = timed(lambda: ops.linear_metrics_walk(vol, 3, 6, rows, walk, beta, sums))
        gb = rows * nv * 4 / 1e9
        print(f"nvars={nv} rows={rows:.3e} volume={gb:.2f} GB: gather {tg:.3f} ms, gram {tgram:.3f}, "
              f"metrics {tmet:.3f} | gram_walk {twg:.3f} ms ({gb / twg * 1e3:.0f} GB/s of volume), "
              f"metrics_walk {twm:.3f} ms ({gb / twm * 1e3:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    sys.exit(main())
