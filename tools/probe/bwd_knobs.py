#!/usr/bin/env python3
"""Time the flat weight-gradient pass (kernel + finalize) against its workgroups-per-CU cap, alone
and alternating with the forward, on one probe build (-DWM_PROBE_ENV).  --build / --run."""
import ctypes, os, statistics, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "weather_model_amd", "csrc")
EXTRA = [a for a in sys.argv if a.startswith("-DWM_")]
OUT = os.path.join(HERE, "variants", "libwm_probe_env" + "".join(e[4:].lower() for e in EXTRA) + ".so")


def build():
    os.makedirs(os.path.join(HERE, "variants"), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-DWM_PROBE_ENV", *EXTRA, "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, f) for f in ("conv2d.hip", "conv3d.hip", "common.hip")]
    subprocess.check_call(cmd + ["-o", OUT])


def run():
    import torch
    N, H, W = 64, 721, 1440
    nset = 3
    xs = [torch.rand(N, 1, H, W, device="cuda") for _ in range(nset)]
    gys = [torch.rand(N, 1, H - 2, W - 2, device="cuda") for _ in range(nset)]
    ys = [torch.empty(N, 1, H - 2, W - 2, device="cuda") for _ in range(nset)]
    w = torch.rand(1, 1, 3, 3, device="cuda"); b = torch.rand(1, device="cuda")
    gw = torch.zeros(10, device="cuda")
    ws = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
    L = ctypes.CDLL(OUT)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.wm_conv2d_fwd_f32.argtypes = [vp] * 4 + [ci] * 8 + [vp]
    L.wm_conv2d_bwd_wb_f32.argtypes = [vp] * 5 + [ctypes.c_size_t] + [ci] * 7 + [vp]
    L.wm_conv2d_bwd_data_f32.argtypes = [vp] * 3 + [ci] * 7 + [vp]
    gxs = [torch.empty(N, 1, H, W, device="cuda") for _ in range(nset)]

    def bwdd(i):
        assert L.wm_conv2d_bwd_data_f32(gys[i].data_ptr(), w.data_ptr(), gxs[i].data_ptr(),
                                        N, 1, H, W, 1, 3, 3, st) == 0
    st = torch.cuda.current_stream().cuda_stream
    mb = 4 * (N * H * W + N * (H - 2) * (W - 2)) / 1e6

    def fwd(i):
        assert L.wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(), ys[i].data_ptr(),
                                   N, 1, H, W, 1, 3, 3, 0, st) == 0

    def bwd(i):
        assert L.wm_conv2d_bwd_wb_f32(xs[i].data_ptr(), gys[i].data_ptr(), gw.data_ptr(),
                                      gw[9:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1,
                                      3, 3, st) == 0

    def burst(fns, reps=20):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(reps + 3):
            if rep == 3: ev[0].record()
            for f in fns: f(rep % nset)
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps * 1e3

    if "--bwdd" in sys.argv:
        for xr, nb in [(8, 2), (16, 2), (8, 3), (16, 3), (16, 4), (16, 1)]:
            os.environ["WM_FWD_XCD_RUN"] = str(xr); os.environ["WM_FWD_BATCHES"] = str(nb)
            bwdd(0); torch.cuda.synchronize()
            ok = torch.allclose(gxs[0][:1], torch.nn.grad.conv2d_input((1, 1, H, W), w, gys[0][:1]), atol=1e-5)
            print(f"bwd_data xcd {xr} batches {nb}: {burst([bwdd]):6.1f} us  fwd+bwd+bwdd {burst([fwd, bwd, bwdd]):6.1f} ok={ok}", flush=True)
        return
    ref = None
    for cap in (2, 3, 4, 6, 8):
        os.environ["WM_BWD_OCC_CAP"] = str(cap)
        bwd(0); torch.cuda.synchronize()
        if ref is None: ref = gw.clone()
        ok = torch.allclose(gw, ref, rtol=1e-5)
        tb = burst([bwd]); tp = burst([fwd, bwd])
        print(f"bwd occ cap {cap}: bwd alone {tb:6.1f} us ({mb / tb * 1e-3:5.2f} TB/s)  fwd+bwd {tp:6.1f} us "
              f"({2 * mb / tp * 1e-3:5.2f} TB/s = {2 * mb / tp * 1e-3 / 8:.3f})  ok={ok}", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv: build()
    if "--run" in sys.argv: run()
