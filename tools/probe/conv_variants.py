#!/usr/bin/env python3
"""Build (--build, CPU) and time (--run, GPU) variants of the stencil kernels: rows per load
batch (WM_STENCIL_RB) x resident blocks per CU cap (WM_STENCIL_BPC).  Interleaved rounds in one
process (cdna_hip_programming.md rule 24)."""
import ctypes, itertools, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VDIR = os.path.join(HERE, "variants")
CSRC = os.path.join(ROOT, "weather_model_amd", "csrc")
RBS = [2, 4]
BPCS = [3, 4, 6]
EXPS = ["", "-DWM_EXP_NTSTORE"]

def build():
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for f in os.listdir(VDIR):
        os.remove(os.path.join(VDIR, f))
    for (rb, bpc), (ei, ex) in itertools.product(itertools.product(RBS, BPCS), enumerate(EXPS)):
        out = os.path.join(VDIR, f"libwm_rb{rb}_bpc{bpc}_e{ei}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               f"-DWM_STENCIL_RB={rb}", f"-DWM_STENCIL_BPC={bpc}", *ex.split(), "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, os.path.join(CSRC, "conv2d.hip"), os.path.join(CSRC, "common.hip"), "-o", out]
        procs.append(subprocess.Popen(cmd))
        if len(procs) >= 6:
            for p in procs: p.wait()
            procs = []
    for p in procs: p.wait()

def run():
    import torch
    N, H, W = 64, 721, 1440
    nset = 3
    xs = [torch.rand(N, 1, H, W, device="cuda") for _ in range(nset)]
    gys = [torch.rand(N, 1, H - 2, W - 2, device="cuda") * 2 - 1 for _ in range(nset)]
    ys = [torch.empty(N, 1, H - 2, W - 2, device="cuda") for _ in range(nset)]
    w = torch.rand(1, 1, 3, 3, device="cuda"); b = torch.rand(1, device="cuda")
    gw = torch.zeros(10, device="cuda")
    ws = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
    libs = {}
    for (rb, bpc), (ei, ex) in itertools.product(itertools.product(RBS, BPCS), enumerate(EXPS)):
        L = ctypes.CDLL(os.path.join(VDIR, f"libwm_rb{rb}_bpc{bpc}_e{ei}.so"))
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.wm_conv2d_fwd_f32.argtypes = [vp] * 4 + [ci] * 8 + [vp]
        L.wm_conv2d_bwd_wb_f32.argtypes = [vp] * 5 + [ctypes.c_size_t] + [ci] * 7 + [vp]
        libs[(rb, bpc, ex)] = L
    st = torch.cuda.current_stream().cuda_stream
    res = {k: ([], []) for k in libs}
    for rnd in range(8):
        for k, L in libs.items():
            i = rnd % nset
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            rc = L.wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(), ys[i].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st)
            e[1].record()
            rc |= L.wm_conv2d_bwd_wb_f32(xs[i].data_ptr(), gys[i].data_ptr(), gw.data_ptr(), gw[9:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3, st)
            e[2].record()
            torch.cuda.synchronize()
            assert rc == 0
            if rnd >= 2:
                res[k][0].append(e[0].elapsed_time(e[1])); res[k][1].append(e[1].elapsed_time(e[2]))
    # steady state of each kernel alone (a kernel also pays for the dirty lines its predecessor
    # left in L2 / Infinity Cache, so A-B-A-B timing mis-attributes the forward's write-back)
    solo = {}
    for k, L in libs.items():
        for which in ("fwd", "bwd"):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for rep in range(12):
                i = rep % nset
                if rep == 2: ev[0].record()
                if which == "fwd":
                    L.wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(), ys[i].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st)
                else:
                    L.wm_conv2d_bwd_wb_f32(xs[i].data_ptr(), gys[i].data_ptr(), gw.data_ptr(), gw[9:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3, st)
            ev[1].record(); torch.cuda.synchronize()
            solo[(k, which)] = ev[0].elapsed_time(ev[1]) / 10
    import statistics
    mb = 4 * (N * H * W + N * (H - 2) * (W - 2)) / 1e6
    for k in sorted(res):
        f, bw = statistics.median(res[k][0]), statistics.median(res[k][1])
        print(f"rb {k[0]} bpc {k[1]} {k[2]:45s}: fwd {f*1e3:7.1f} us {mb/f:6.0f} GB/s | bwd_wb {bw*1e3:7.1f} us {mb/bw:6.0f} GB/s | sum {(f+bw)*1e3:6.1f} | solo fwd {solo[(k,'fwd')]*1e3:6.1f} bwd {solo[(k,'bwd')]*1e3:6.1f}")

if __name__ == "__main__":
    if "--build" in sys.argv: build()
    if "--run" in sys.argv: run()
