#!/usr/bin/env python3
"""Sweep the forward flat stencil's launch knobs (ranges per XCD run, load batches per range,
occupancy cap through the dynamic-LDS floor) in ONE process on one probe build (-DWM_PROBE_ENV reads
the knobs from the environment at every call).  --build on the CPU, --run on the GPU."""
import ctypes, itertools, os, statistics, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "weather_model_amd", "csrc")
RBS = [int(a[4:]) for a in sys.argv if a.startswith("--rb")] or [4]
EXTRA = [a[2:] for a in sys.argv if a.startswith("-DWM_")]


def out(rb):
    tag = "".join(e.replace("WM_", "_").lower() for e in EXTRA)
    return os.path.join(HERE, "variants", f"libwm_probe_env_rb{rb}{tag}.so")


def build():
    os.makedirs(os.path.join(HERE, "variants"), exist_ok=True)
    procs = []
    for rb in RBS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-DWM_PROBE_ENV", f"-DWM_FWD_RB={rb}", *["-D" + e for e in EXTRA], "-I", os.path.join(ROOT, "include"), "-I", CSRC]
        cmd += [os.path.join(CSRC, f) for f in ("conv2d.hip", "conv3d.hip", "common.hip")]
        procs.append(subprocess.Popen(cmd + ["-o", out(rb)]))
    assert all(p.wait() == 0 for p in procs)


def run():
    import torch
    N, H, W = 64, 721, 1440
    nset = 3
    xs = [torch.rand(N, 1, H, W, device="cuda") for _ in range(nset)]
    ys = [torch.empty(N, 1, H - 2, W - 2, device="cuda") for _ in range(nset)]
    w = torch.rand(1, 1, 3, 3, device="cuda"); b = torch.rand(1, device="cuda")
    vp, ci = ctypes.c_void_p, ctypes.c_int
    libs = {}
    for rb in RBS:
        libs[rb] = ctypes.CDLL(out(rb))
        libs[rb].wm_conv2d_fwd_f32.argtypes = [vp] * 4 + [ci] * 8 + [vp]
    cur = {"L": libs[RBS[0]]}
    st = torch.cuda.current_stream().cuda_stream
    mb = 4 * (N * H * W + N * (H - 2) * (W - 2)) / 1e6
    ref = torch.nn.functional.conv2d(xs[0][:2], w, b)

    def fwd(i, kh=3, kw=3):
        rc = cur["L"].wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(), ys[i].data_ptr(),
                                 N, 1, H, W, 1, kh, kw, 0, st)
        assert rc == 0

    def timeit(fwd, reps=15):
        ts = []
        for rep in range(reps + 3):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record(); fwd(rep % nset); e[1].record(); torch.cuda.synchronize()
            if rep >= 3:
                ts.append(e[0].elapsed_time(e[1]) * 1e3)
        return statistics.median(ts), min(ts)

    def burst(fwd, reps=12):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(reps + 2):
            if rep == 2: ev[0].record()
            fwd(rep % nset)
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps * 1e3

    def sustained(fwd, lead=60, reps=60):
        """average over `reps` launches that follow `lead` back-to-back launches (~6 ms of load):
        the chip slows down after ~2 ms of sustained streaming (clock management), and that state
        is the one a training loop runs in"""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(lead + reps):
            if rep == lead: ev[0].record()
            fwd(rep % nset)
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps * 1e3

    combos = list(itertools.product(RBS, [2, 4, 8], [1, 2, 3], [0, 32768, 40960, 53248, 81920]))
    if "--fine" in sys.argv:
        combos = list(itertools.product(RBS, [8], [1, 2, 3, 4, 6], [0, 40960, 53248, 81920]))
    if "--fine2" in sys.argv:
        combos = list(itertools.product(RBS, [16], [1, 2, 3, 4], [0, 40960]))
    shapes = [(3, 3)]
    if "--shapes" in sys.argv:
        combos = [(2, 8, 3, 0), (2, 8, 2, 0)]
        shapes = [(1, 1), (3, 1), (1, 3), (2, 2), (3, 3)]
    for (rb, xr, nb, lds), (kh, kw) in itertools.product(combos, shapes):
        cur["L"] = libs[rb]
        w = torch.rand(1, 1, kh, kw, device="cuda")
        ys = [torch.empty(N, 1, H - kh + 1, W - kw + 1, device="cuda") for _ in range(nset)]
        ref = torch.nn.functional.conv2d(xs[0][:2], w, b)
        mb = 4 * (N * H * W + N * (H - kh + 1) * (W - kw + 1)) / 1e6

        def fwd(i, kh=kh, kw=kw, w=w, ys=ys):
            rc = cur["L"].wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(),
                                            ys[i].data_ptr(), N, 1, H, W, 1, kh, kw, 0, st)
            assert rc == 0
        os.environ["WM_FWD_XCD_RUN"] = str(xr)
        os.environ["WM_FWD_BATCHES"] = str(nb)
        os.environ["WM_FWD_LDS_MIN"] = str(lds)
        fwd(0); torch.cuda.synchronize()
        ok = torch.allclose(ys[0][:2], ref, atol=1e-5)
        med, mn = timeit(fwd); bu = burst(fwd); su = sustained(fwd); time.sleep(0.05)
        print(f"({kh},{kw}) rb {rb} xcd_run {xr:2d} batches {nb} lds_min {lds:6d}: median {med:6.1f} us min {mn:6.1f} "
              f"burst {bu:6.1f} us  {mb / bu * 1e3 / 1e3:5.2f} TB/s  sustained {su:6.1f}  ok={ok}", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv: build()
    if "--run" in sys.argv: run()
