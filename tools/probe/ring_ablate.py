#!/usr/bin/env python3
"""Time conv_ring_kernel (conv32 shape: 8 x 32 x 721 x 1440, 3x3) forward and input gradient on
probe builds with parts of the kernel removed (-DRG_ABL_NO_MMA / NO_FILL / NO_STORE / NO_LDS_B ...).
  ring_ablate.py --build -DRG_ABL_NO_MMA     (here)
  ring_ablate.py --run   -DRG_ABL_NO_MMA     (GPU box)
Outputs of an ablated build are wrong on purpose; the plain build is checked against torch."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "weather_model_amd", "csrc")
EXTRA = [a for a in sys.argv if a.startswith("-DRG_") or a.startswith("-DWM_")]
OUT = os.path.join(HERE, "variants", "libwm_ring" + "".join("_" + e[5:].lower() for e in EXTRA) + ".so")


def build():
    os.makedirs(os.path.join(HERE, "variants"), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-DWM_PROBE_ENV", *EXTRA, "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, f) for f in ("conv2d.hip", "conv3d.hip", "conv_ring.hip", "common.hip")]
    subprocess.check_call(cmd + ["-o", OUT], stderr=subprocess.DEVNULL)


def run():
    import torch
    N, C = 8, 32
    H, W = int(os.environ.get('RG_H', 721)), int(os.environ.get('RG_W', 1440))
    RELU = int(os.environ.get('RG_RELU', 0))
    x = torch.rand(N, C, H, W, device="cuda") - 0.5
    w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) / 17
    b = torch.rand(C, device="cuda") - 0.5
    y = torch.empty(N, C, H - 2, W - 2, device="cuda")
    gy = torch.rand(N, C, H - 2, W - 2, device="cuda") - 0.5
    gx = torch.empty_like(x)
    L = ctypes.CDLL(OUT)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.wm_conv2d_fwd_f32.argtypes = [vp] * 4 + [ci] * 8 + [vp]
    L.wm_conv2d_bwd_data_f32.argtypes = [vp] * 3 + [ci] * 7 + [vp]
    L.wm_conv2d_bwd_data_relu_f32.argtypes = [vp] * 4 + [ci] * 7 + [vp]
    mask = torch.rand(N, C, H, W, device="cuda") - 0.5
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        assert L.wm_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, W, C, 3, 3, RELU, st) == 0

    def bwd():
        assert L.wm_conv2d_bwd_data_f32(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), N, C, H, W, C, 3, 3, st) == 0

    def bwdm():
        assert L.wm_conv2d_bwd_data_relu_f32(gy.data_ptr(), w.data_ptr(), mask.data_ptr(), gx.data_ptr(), N, C, H, W, C, 3, 3, st) == 0

    def burst(fn, reps=int(os.environ.get("RG_REPS", "10")), pre=int(os.environ.get("RG_PRE", "40"))):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(reps + pre):
            if rep == pre: ev[0].record()
            fn()
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    flops = 2.0 * C * C * 9
    tm = burst(bwdm)
    tf, tb = burst(fwd), burst(bwd)
    msg = f"{os.path.basename(OUT):40s} fwd {tf:6.3f} ms {flops * N * (H - 2) * (W - 2) / tf / 1e9:6.1f} TF   bwd_data {tb:6.3f} ms {flops * N * H * W / tb / 1e9:6.1f} TF  masked {tm:6.3f} ms"
    if not EXTRA or all(e.startswith("-DWM_") or e.startswith("-DRG_OPT") for e in EXTRA):
        ref = torch.nn.functional.conv2d(x[:1], w, b)
        if RELU: ref = ref.clamp_min(0)
        e1 = float((y[:1] - ref).abs().max() / ref.abs().max())
        refg = torch.nn.grad.conv2d_input((1, C, H, W), w, gy[:1])
        e2 = float((gx[:1] - refg).abs().max() / refg.abs().max())
        bwdm(); torch.cuda.synchronize()
        e3 = float((gx[:1] - refg * (mask[:1] > 0)).abs().max() / refg.abs().max())
        msg += f" masked {e3:.1e}"
        msg += f"  relerr fwd {e1:.1e} bwd {e2:.1e}"
    print(msg, flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv: build()
    if "--run" in sys.argv: run()
