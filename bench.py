#!/usr/bin/env python3
"""bench.py -- headline benchmark of the weather-model hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|gram16|conv3x3|gram64|fit_walk|...]

One JSON line on stdout (rank 0).  Primary workload (`value`) = BASELINE.json configs[1]:
16-feature least squares on 1e8 synthetic rows per GPU (Gram build + normal-equation solve),
rows resident in HBM before the timed region.  A "step" = zero the accumulator, one pass of the
Gram kernel over all rows, (N>1: one RCCL all-reduce of the 18x18 float64 partial Gram), one
solve.  `also.conv3x3` carries BASELINE configs[2] (3x3 single-channel conv forward +
backward(weight,bias) over 64 x 721 x 1440 grids) measured in the same run, with its own roofline.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), rows / images sharded
with no data-path collective; the only exchange is the partial-Gram (or 10-float gradient)
all-reduce.  Weak scaling: per-GPU work is fixed.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: FP32 matrix 157.3 TFLOP/s
EXEC_FLOPS_PER_SAMPLE = 101.6e6 * 2048 / 1038240    # model_kernel<1>: MFMAs issued per step


PMC_FILE = os.path.join("profiles", "r03", "pmc_traffic.json")
PMC_FALLBACKS = (os.path.join("profiles", "r02", "pmc_traffic.json"),
                 os.path.join("profiles", "r01", "pmc_traffic.json"))


def pmc_traffic(kernels, default_shape, path=None):
    """HBM-side bytes per launch of `kernels` from the committed PMC passes (tools/pmc_bench.sh +
    tools/pmc_summarise.py, FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes).  The
    counters cannot be collected inside a timed run, so this is the figure of the STORED profile
    of this same command (`traffic_source` names the file), not a live counter; it is only reported
    when the workload has the profiled (default) shape."""
    if not default_shape:
        return None
    for path in ((path,) if path else (PMC_FILE,) + PMC_FALLBACKS):
        try:
            with open(os.path.join(ROOT, path)) as fh:
                table = json.load(fh)["kernels"]
            return float(sum(table[k]["hbm_bytes"] for k in kernels))
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_source(kernels, default_shape, path=None):
    if not default_shape:
        return None
    for path in ((path,) if path else (PMC_FILE,) + PMC_FALLBACKS):
        try:
            with open(os.path.join(ROOT, path)) as fh:
                table = json.load(fh)["kernels"]
            if all(k in table for k in kernels):
                return "stored profile " + path
        except (OSError, KeyError, ValueError):
            continue
    return None


def visible_gpus(base="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process may use, counted WITHOUT touching HIP: torch.cuda.device_count() falls back
    to hipGetDeviceCount when amdsmi is not importable, which initialises the runtime in the
    launcher.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES if set, else the
    KFD topology: nodes with simd_count > 0 are GPUs (CPUs have simd_count 0)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as fh:
                    props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return None                 # no KFD here: let the ranks fail loudly instead
    return n


def launch_ranks(ngpus, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process, which never
    touches the GPU (devices are counted from the environment or the KFD topology files, not
    through HIP), starts N fresh child processes, one per device, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and exits with the worst child status.
    Nothing is exec'ed from a process that has initialised HIP.  PROFILING IS DONE PER RANK: put
    the single-rank program directly after `rocprofv3 ... --` (with RANK/WORLD_SIZE/MASTER_* in the
    environment for N > 1), never this self-launcher -- the profiler's preloaded library
    initialises the GPU in the launcher, and its children would then be started from a
    GPU-initialised process."""
    import socket
    import subprocess
    rehearsal = bool(os.environ.get("WM_BENCH_REHEARSAL"))
    ndev = visible_gpus()
    if ndev is not None and ndev < ngpus and not rehearsal and "plumbing" not in argv:
        raise SystemExit(f"bench.py --gpus {ngpus}: only {ndev} GPU(s) visible "
                         f"(WM_BENCH_REHEARSAL=1 runs all ranks on GPU 0 over gloo)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus),
                   LOCAL_WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WM_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    for q in procs[1:]:
        try:
            codes.append(q.wait(timeout=120))
        except subprocess.TimeoutExpired:
            q.kill()                      # exactly the PID this launcher started
            codes.append(q.wait())
    # rank 0's stdout may carry library chatter (gloo prints its connection line there): the JSON
    # line goes to stdout alone, everything else to stderr
    lines = out0.splitlines()
    js = [k for k, ln in enumerate(lines) if ln.startswith("{") and ln.rstrip().endswith("}")]
    for k, ln in enumerate(lines):
        if not js or k != js[-1]:
            print(ln, file=sys.stderr)
    if js:
        print(lines[js[-1]])
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    if bad:
        raise SystemExit(f"bench.py launcher: rank exit codes {codes}")


def dist_setup(ngpus, cpu_only=False):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    info = {"rccl_ranks": 1, "backend": None,
            "launcher": os.environ.get("WM_BENCH_LAUNCHER",
                                       "torchrun" if world > 1 else "single")}
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # WM_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- exercises the N > 1 code path on
        # a one-GPU box (numbers from such a run mean nothing)
        if cpu_only:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        elif os.environ.get("WM_BENCH_REHEARSAL"):
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        info["rccl_ranks"] = dist.get_world_size()
        info["backend"] = dist.get_backend()
        if info["backend"] == "gloo" and not cpu_only:
            info["rehearsal"] = "all ranks on GPU 0 over gloo: rates mean nothing"
    elif not cpu_only:
        torch.cuda.set_device(0)
    if world != ngpus and rank == 0:
        print(f"[bench] note: --gpus {ngpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    return rank, world, local, info


def cpu_threads():
    """Threads the CPU baselines may use: the box's CPU share (OMP_NUM_THREADS is set to it on the
    GPU box; os.cpu_count() reports the whole machine there)."""
    n = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS", "")
    return max(1, min(n, int(env))) if env.isdigit() else n


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def median_time(fn, budget_s, min_reps=3, max_reps=20):
    """Median wall time of fn() over at least min_reps and at most max_reps runs within budget_s."""
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < min_reps or (time.perf_counter() < t_end and len(ts) < max_reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value, world):
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64,
                     device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


PREWARM_MS = float(os.environ.get("WM_BENCH_PREWARM_MS", "100"))
PREWARM_LOG = {}


def prewarm(step, world, cuda, tag):
    """Untimed setup before the W warm-ups: run the workload's own step for ~PREWARM_MS of GPU time.
    A chip that has been idle runs its first tens of milliseconds below its steady clocks (the
    stencil pair: 190 us cold, 178 us after 70 ms of the same launches; profiles/r03), and W = 5
    steps of a 0.2 ms step do not get past that.  The count is agreed over ranks (steps may hold
    collectives).  Recorded per workload as `prewarm_steps`."""
    if PREWARM_MS <= 0:
        PREWARM_LOG[tag] = 0
        return
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    step(0, False)
    sync()
    t0 = time.perf_counter()
    step(1, False)
    step(2, False)
    sync()
    est = max((time.perf_counter() - t0) / 2, 1e-6)
    n = int(max_over_ranks(float(min(2000, max(0, int(PREWARM_MS * 1e-3 / est)))), world))
    for i in range(n):
        step(3 + i, False)
    sync()
    PREWARM_LOG[tag] = n + 3


def timed_steps(step, steps, warmup, world, cuda=True, tag=None):
    """(setup: prewarm) W untimed warm-ups, then exactly K steps between barrier+synchronize pairs."""
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    if tag is not None:
        prewarm(step, world, cuda, tag)
    for i in range(warmup):
        step(i, False)
    sync()
    barrier(world)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i, True)
    sync()
    barrier(world)
    t1 = time.perf_counter()
    return max_over_ranks(t1 - t0, world)


class Exchange:
    """The data-parallel exchange step of a workload, with HIP events around it on every fourth
    timed step (`exchange_us` in the JSON line: what the all-reduce costs on the stream, so that a
    scaling run explains its own efficiency).  No-op at N = 1."""

    def __init__(self, world):
        self.world = world
        self.ev = []
        self.calls = 0

    def run(self, fn, timed=True):
        if self.world == 1:
            return
        if timed:
            self.calls += 1
        if timed and self.calls % 4 == 1:
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e[0].record()
            fn()
            e[1].record()
            self.ev.append(e)
        else:
            fn()

    def us(self):
        if not self.ev:
            return None
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in self.ev])) * 1e3


# ------------------------------------------------------------------------------ regression
def gram_cpu_baselines(X, Y, F, P, beta_gpu):
    """CPU legs beside the Gram numbers (rank 0, N = 1): `cpu_baseline` = the library path the
    contract names (BASELINE.md 3 / SURVEY 8d: numpy float32 A.T @ A, A.T @ y with A = [X | 1], then
    a float64 lstsq of the normal equations), `cpu_baseline_port` = the repo's own C/OpenMP oracle."""
    ns = 4_000_000 if F <= 16 else (1_000_000 if F <= 64 else 100_000)
    ns = min(ns, X.shape[0])
    Xs, Ys = X[:ns].cpu().numpy(), Y[:ns].cpu().numpy()
    out = {}
    A = np.concatenate([Xs, np.ones((ns, 1), np.float32)], axis=1)
    sol = [None]

    def lib():
        G = A.T @ A
        c = A.T @ Ys
        sol[0] = np.linalg.lstsq(G.astype(np.float64), c.astype(np.float64), rcond=None)[0]

    med, reps = median_time(lib, 8.0, min_reps=3, max_reps=40)
    out["cpu_baseline"] = {
        "value": ns / med, "unit": "rows/s", "cores": cpu_threads(), "kind": "library",
        "cpu": cpu_model_name(),
        "sample": f"{ns} of the same rows (F={F}): numpy float32 A.T@A, A.T@y (A=[X|1]) + float64 "
                  f"lstsq of the normal equations, BLAS threads, median of {reps}",
        "coef_max_diff_vs_gpu": float(np.abs(sol[0][:F] - beta_gpu[:F]).max()),
        "coef_diff_note": "the CPU fit sees only the sample, the GPU fit all rows"}
    from oracle import cpu as ocpu
    bsol = [None]

    def port():
        bsol[0] = ocpu.normal_solve(ocpu.gram(Xs, Ys, f32=True), F, P)

    medp, repsp = median_time(port, 4.0, min_reps=2, max_reps=20)
    out["cpu_baseline_port"] = {
        "value": ns / medp, "unit": "rows/s", "cores": ocpu.threads(), "kind": "port",
        "sample": f"{ns} of the same rows, oracle/wm_oracle.c float32 Gram + float64 solve, OpenMP, "
                  f"median of {repsp}",
        "coef_max_diff_vs_gpu": float(np.abs(bsol[0] - beta_gpu).max())}
    return out


def gram_kernel_name(F, P):
    if (F, P) == (16, 1):
        return "gram_f16_kernel"
    if (F, P) == (64, 1):
        return "gram_x64_kernel"
    if F + P <= 15:
        return "gram_small_kernel"
    if (F, P) in ((16, 8), (16, 4), (8, 8)):
        return "gram_quad_kernel"
    if (P + 3) // 4 <= 3 and (F + 3) // 4 + (P + 3) // 4 <= 9:
        return "gram_quadu_kernel"
    return "gram_mid_kernel" if F + P + 1 <= 64 else "gram_syrk_kernel"


def run_gram_family(rank, world, ps=(3, 5, 6, 7, 10, 12), bytes_target=4.8e9):
    """The rest of multipleRegression's family F = 2p, P = p (p variables at CUBE_SIZE 1,
    multipleRegression.py:273-286): the Gram pass and the loss / mae / msle pass of an epoch on
    resident rows, each bracketed by HIP events -- short (about 4.8 GB of rows per shape, six timed
    launches), so that the default line carries a measured figure for every kernel family the fit can
    take (narrow-row stream with dense loads, padded quad MFMA, narrow metrics, tile metrics)."""
    from weather_model_amd import ops, synth, _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for p_ in ps:
        F, P = 2 * p_, p_
        n = int(bytes_target / (4 * (F + P))) // 4096 * 4096
        X = torch.empty((n, F), device="cuda", dtype=torch.float32)
        Y = torch.empty((n, P), device="cuda", dtype=torch.float32)
        bt = torch.from_numpy(synth.normal(1001, (F, P))).cuda()
        _lib.check(L.wm_synth_uniform_f32(X.data_ptr(), n * F, 1, rank * n * F, 0.0, 1.0, st))
        _lib.check(L.wm_synth_uniform_f32(Y.data_ptr(), n * P, 2001, rank * n * P,
                                          -0.01 * 3 ** 0.5, 0.01 * 3 ** 0.5, st))
        for s0 in range(0, n, 250_000):
            Y[s0:s0 + 250_000].add_(X[s0:s0 + 250_000] @ bt + 0.5)
        plan = ops.GramPlan(n, F, P, "cuda")
        sums = torch.zeros((4,), device="cuda", dtype=torch.float64)
        plan.acc.zero_()
        plan.accumulate(X, Y)
        plan.solve()
        gev, mev = [], []
        for i in range(8):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            plan.acc.zero_()
            e[0].record()
            plan.accumulate(X, Y)
            e[1].record()
            sums.zero_()
            e[2].record()
            ops.linear_metrics(X, Y, plan.beta, sums)
            e[3].record()
            if i >= 2:
                gev.append((e[0], e[1]))
                mev.append((e[2], e[3]))
        torch.cuda.synchronize()
        gms = float(np.mean([a.elapsed_time(b) for a, b in gev]))
        mms = float(np.mean([a.elapsed_time(b) for a, b in mev]))
        gb = 4.0 * (F + P) * n / 1e6
        out[f"F{F}_P{P}"] = {
            "rows": n, "gram_ms": gms, "gram_GBs": gb / gms, "gram_frac": gb / gms / HBM_PEAK_GBS,
            "gram_kernel": gram_kernel_name(F, P), "metrics_ms": mms, "metrics_GBs": gb / mms,
            "metrics_frac": gb / mms / HBM_PEAK_GBS, "mse": float(sums[0] / sums[3]),
            "planted_coef_max_err": float((plan.beta[:F].float() - bt).abs().max())}
        del X, Y, plan
        torch.cuda.empty_cache()
    return {"unit": "GB/s of 4(F+P) algorithmic bytes per row (HBM-bound: F/2 flop/B < 19.7)",
            "config": {"workload": "multipleRegression's family F = 2p, P = p for p = "
                                   + ", ".join(str(v) for v in ps)
                                   + ": Gram pass and loss/mae/msle pass on resident rows, HIP events, "
                                     "6 timed launches each"},
            "shapes": out}


def run_fit_walk(rank, world, nvars=(1, 4), files=1200, grid=(16, 64, 64), bs=16):
    """An epoch of multipleRegression's own fit over a RESIDENT volume (CUBE_SIZE 1, BATCH_SIZE 16,
    TIME_SPAN 6: multipleRegression.py:29,35,40,217-270), both ways, HIP events around each pass:
    rows gathered by wm_gather_cubes_f32, then wm_gram_f32 and wm_linear_metrics_f32 -- against
    wm_gram_walk_f32 and wm_linear_metrics_walk_f32, which read the volume themselves.  The volume is
    stored point-major (Z, Y, X, files, vars) as the drop-in keeps it; one epoch = one visit of every
    grid cell (length batches each)."""
    from weather_model_amd import ops, _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    Z, Y, X = grid
    out = {}
    for nv in nvars:
        stored = torch.empty((Z, Y, X, files, nv), device="cuda", dtype=torch.float32)
        _lib.check(L.wm_synth_uniform_f32(stored.data_ptr(), stored.numel(), 77 + nv,
                                          rank * stored.numel(), 0.0, 1.0, st))
        vol = stored.permute(3, 4, 0, 1, 2)
        length = (files - 6) // bs - 1
        walk = {"batch": bs, "length": length, "index": 0, "region": (0, 0, 0), "dims": grid}
        rows = bs * length * Z * Y * X
        D = 3 * nv + 1
        acc = torch.zeros((D, D), device="cuda", dtype=torch.float64)
        sums = torch.zeros((4,), device="cuda", dtype=torch.float64)
        ops.gram_walk(vol, 3, 6, rows, walk, acc)
        beta = ops.normal_solve(acc, 2 * nv, nv)
        t = {k: [] for k in ("gather", "gram", "metrics", "gram_walk", "metrics_walk")}
        for i in range(8):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
            ev[0].record()
            Xr, Yr = ops.gather_cubes(vol, (1, 1, 1), 3, 6, rows, walk=walk)
            ev[1].record()
            ops.gram(Xr, Yr, acc)
            ev[2].record()
            ops.linear_metrics(Xr, Yr, beta, sums)
            ev[3].record()
            del Xr, Yr
            ev[4].record()
            ops.gram_walk(vol, 3, 6, rows, walk, acc)
            ev[5].record()
            ops.linear_metrics_walk(vol, 3, 6, rows, walk, beta, sums)
            ev[6].record()
            torch.cuda.synchronize()
            if i >= 2:
                for k, (a, b) in zip(t, ((0, 1), (1, 2), (2, 3), (4, 5), (5, 6))):
                    t[k].append(ev[a].elapsed_time(ev[b]))
        ms = {k: float(np.mean(v)) for k, v in t.items()}
        gb = 4.0 * nv * rows / 1e6                    # every value of the walked files read once
        rows_ms = ms["gather"] + ms["gram"] + ms["metrics"]
        walk_ms = ms["gram_walk"] + ms["metrics_walk"]
        out[f"V{nv}"] = {
            "rows": rows, "F": 2 * nv, "P": nv, "ms": ms, "epoch_ms_rows": rows_ms, "epoch_ms_walk": walk_ms,
            "speedup": rows_ms / walk_ms, "rows_per_s_walk": rows / walk_ms * 1e3 * world,
            "gram_walk_GBs": gb / ms["gram_walk"], "gram_walk_frac": gb / ms["gram_walk"] / HBM_PEAK_GBS,
            "metrics_walk_GBs": gb / ms["metrics_walk"],
            "bound": "vector ALU (float32 products per row), not HBM: 4 V bytes per row"}
        del stored, vol
        torch.cuda.empty_cache()
    return {"unit": "ms per pass over one epoch's rows; GB/s of 4 V algorithmic bytes per row",
            "config": {"workload": f"multipleRegression epoch over a resident volume: {files} files, grid "
                                   f"{Z}x{Y}x{X}, BATCH_SIZE {bs}, CUBE_SIZE 1, V variables (F = 2V, P = V), "
                                   f"point-major volume; gather + Gram + metrics against the passes that "
                                   f"read the volume directly, 6 timed launches each"},
            "shapes": out}


def run_gram(F, n_rows, steps, warmup, rank, world, cpu_base, P=1, dist="uniform"):
    """Least squares with F predictors and P targets on n_rows resident rows per GPU: one step =
    zero the accumulator, Gram pass, (N > 1: all-reduce of the (F+1+P)^2 float64 Gram), solve.
    The roofline is the Gram kernel's: HBM (4(F+P) B/row) while F/2 flop/B is below the fp32 MFMA
    ridge of 19.7, MFMA (2F(F+P) flop/row) above it."""
    from weather_model_amd import ops, synth, dp
    beta_true = synth.normal(1001, (F, P))
    # rows of this rank: global row index keeps the data independent of the GPU count
    row_off = rank * n_rows
    X = torch.empty((n_rows, F), device="cuda", dtype=torch.float32)
    Y = torch.empty((n_rows, P), device="cuda", dtype=torch.float32)
    bt = torch.from_numpy(beta_true).cuda()
    from weather_model_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    if P == 1:
        slab = 25_000_000          # generate in slabs to bound the transient footprint
        for s0 in range(0, n_rows, slab):
            m = min(slab, n_rows - s0)
            _lib.check(L.wm_synth_regression_f32(X[s0:].data_ptr(), Y[s0:].data_ptr(), m, F, P, 1,
                                                 row_off + s0, bt.data_ptr(), 0.5, 0.01,
                                                 1 if dist == "normal" else 0, st))
    else:
        # many targets: the per-row generator would hash every feature once per target; the rows
        # come from the same counter-based stream, the targets from one library product (set-up
        # only, outside every timed region), noise uniform with standard deviation 0.01
        _lib.check(L.wm_synth_uniform_f32(X.data_ptr(), n_rows * F, 1, row_off * F, 0.0, 1.0, st))
        _lib.check(L.wm_synth_uniform_f32(Y.data_ptr(), n_rows * P, 2001, row_off * P,
                                          -0.01 * 3 ** 0.5, 0.01 * 3 ** 0.5, st))
        slab = 250_000
        for s0 in range(0, n_rows, slab):
            Y[s0:s0 + slab].add_(X[s0:s0 + slab] @ bt + 0.5)
    torch.cuda.synchronize()
    plan = ops.GramPlan(n_rows, F, P, "cuda")

    evs, sevs = [], []
    xch = Exchange(world)

    def step(i, timed):
        plan.acc.zero_()
        # the Gram pass is bracketed by HIP events on every other timed step (an event record
        # costs a few microseconds on the stream)
        timed = timed and (steps < 4 or i % 2 == 0)
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        plan.accumulate(X, Y)
        if timed:
            e[1].record()
        xch.run(lambda: dp.reduce_gram_(plan.acc), timed)   # the one exchange step: (F+1+P)^2 float64
        plan.solve()
        if timed:
            if world == 1:
                e[2].record()
                sevs.append((e[1], e[2]))
            evs.append((e[0], e[1]))

    total = timed_steps(step, steps, warmup, world, tag="gram")
    kern_ms = [a.elapsed_time(c) for a, c in evs]
    beta = plan.beta.cpu().numpy()
    err = float(np.abs(beta[:F] - beta_true).max())
    rows_total = n_rows * world * steps
    bytes_per_row = 4 * (F + P)
    flops_per_row = 2 * F * (F + P)
    kavg = float(np.mean(kern_ms)) * 1e-3
    res = {
        "prewarm_steps": PREWARM_LOG.get("gram"),
        "value": rows_total / total,
        "ms_per_step": total / steps * 1e3,
        "planted_coef_max_err": err,
        "kernel_ms_avg": kavg * 1e3,
        "exchange_us": xch.us(),
    }
    if sevs:
        res["solve_ms_avg"] = float(np.mean([a.elapsed_time(c) for a, c in sevs]))
    kname = gram_kernel_name(F, P)
    # the shapes whose launches are in the stored PMC passes of the default line
    default_shape = (F, P, n_rows) in ((16, 1, 100_000_000), (2, 1, 400_000_000), (432, 216, 1_000_000),
                                       (16, 8, 100_000_000))
    nq = 2 if F + P + 1 <= 32 else 4
    pkey = {"gram_small_kernel": f"gram_small_kernel<{F}, {P}>",
            "gram_mid_kernel": f"gram_mid_kernel<{nq}, {'true' if F % nq == 0 and P % nq == 0 else 'false'}>",
            "gram_quad_kernel": f"gram_quad_kernel<{F // 4}, {P // 4}>",
            "gram_syrk_kernel": "gram_syrk_kernel<true, 2>"}.get(kname, kname)
    gbs = bytes_per_row * n_rows / kavg / 1e9
    tf = flops_per_row * n_rows / kavg / 1e12
    if flops_per_row / bytes_per_row < MFMA_F32_PEAK_TF * 1e3 / HBM_PEAK_GBS:
        res["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS}
    else:
        res["roofline"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                           "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                           "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
    res["roofline"].update({"traffic": pmc_traffic([pkey], default_shape),
                            "traffic_source": pmc_source([pkey], default_shape),
                            "kernel": kname, "algorithmic_bytes_per_row": bytes_per_row,
                            "algorithmic_flops_per_row": flops_per_row,
                            "TFLOPs_algorithmic": tf})
    if P > 1 or F == 2:
        # the other half of an epoch of the drop-ins' fit: loss / mae / msle of the solved
        # coefficients over the same resident rows (multipleRegression.py:292-296), one fused pass
        sums = torch.zeros((4,), device="cuda", dtype=torch.float64)
        mev = []
        for i in range(3 + min(steps, 10)):
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e[0].record()
            ops.linear_metrics(X, Y, plan.beta, sums.zero_())
            e[1].record()
            if i >= 3:
                mev.append(e)
        torch.cuda.synchronize()
        mms = float(np.mean([a.elapsed_time(c) for a, c in mev]))
        fused = bool(L.wm_linear_metrics_supported(F, P))
        wide = not fused and bool(L.wm_linear_metrics_wide_supported(F, P))
        mkey = {(16, 8): "linear_metrics_tile_kernel<4, 8, true>",
                (2, 1): "linear_metrics_small_kernel<2, 1, true>"}.get(
                    (F, P), "linear_metrics_wide_kernel<14, true, true, false, false>" if wide
                    else "residual_sums_kernel")
        mgbs = bytes_per_row * n_rows / mms / 1e6
        res["fit_metrics"] = {
            "ms": mms, "mse": float(sums[0] / sums[3]), "mae": float(sums[1] / sums[3]),
            "msle": float(sums[2] / sums[3]), "epoch_ms": total / steps * 1e3 + mms,
            "roofline": {"bound": "hbm", "achieved": mgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": mgbs / HBM_PEAK_GBS, "traffic": pmc_traffic([mkey], default_shape),
                         "traffic_source": pmc_source([mkey], default_shape),
                         "kernel": mkey,
                         "algorithmic_bytes_per_row": bytes_per_row} if fused else
                        {"bound": "mfma", "achieved": 2.0 * F * P * n_rows / mms / 1e9,
                         "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": 2.0 * F * P * n_rows / mms / 1e9 / MFMA_F32_PEAK_TF,
                         "traffic": pmc_traffic([mkey], default_shape),
                         "traffic_source": pmc_source([mkey], default_shape),
                         **({"kernel": mkey + " (predictions on the fp32 MFMA, scored from the "
                                              "accumulators; round 3a: library GEMM + residual pass)"}
                            if wide else
                            {"traffic_note": "residual_sums_kernel only (the GEMM is the library's)",
                             "kernel": "library GEMM (predictions) + residual_sums_kernel"}),
                         "algorithmic_flops_per_row": 2 * F * P}}
    if cpu_base and rank == 0 and world == 1:
        res.update(gram_cpu_baselines(X, Y, F, P, beta))
    del X, Y
    torch.cuda.empty_cache()
    return res


def gram_config(F, P, rows, world, dist, what):
    return {"workload": f"{what}: F={F}, P={P}, {rows} rows per GPU, "
                        f"X~{'N(0,1)' if dist == 'normal' else 'U(0,1)'}, Gram (fp32, fp64 combine) "
                        f"+ normal-equation solve",
            "rows_per_gpu": rows, "features": F, "targets": P,
            "parallelism": f"dp{world} (row shards, one {F + 1 + P}x{F + 1 + P} f64 all-reduce)"}


def run_gram64(steps, warmup, rank, world, cpu_base, rows_per_gpu, chunk_rows=50_000_000):
    """BASELINE configs[4] (config 5 of SURVEY 8d): F = 64 least squares over rows that do NOT fit
    in HBM (1.25e9 rows x 260 B = 325 GB per GPU at the full size).  One step = one pass over all
    `rows_per_gpu` rows of this rank in resident chunks: the counter-based generator writes chunk
    c (keyed on the GLOBAL row index, so any GPU count sees the same data) into one of two buffers,
    wm_gram_f32 adds its Gram to the running float64 accumulator; then (N > 1) one RCCL all-reduce
    of the 66x66 float64 Gram and the solve.  Reported: end-to-end rows/s (generation included) and
    the Gram kernel's own rows/s from HIP events around every wm_gram_f32 call (generation
    excluded)."""
    from weather_model_amd import ops, synth, dp, _lib
    F, P = 64, 1
    L = _lib.lib()
    beta_true = synth.normal(1001, (F, P))
    bt = torch.from_numpy(beta_true).cuda()
    chunk = int(min(chunk_rows, rows_per_gpu))
    nchunks = (rows_per_gpu + chunk - 1) // chunk
    bufs = [(torch.empty((chunk, F), device="cuda"), torch.empty((chunk, P), device="cuda"))
            for _ in range(2)]
    plan = ops.GramPlan(chunk, F, P, "cuda")
    row_off = rank * rows_per_gpu
    pending = []
    xch = Exchange(world)

    # (generating chunk c + 1 on a second stream while chunk c's Gram runs was tried: both kernels are
    # persistent and fill the chip, and the pass got slower -- 148.8 ms against 128.3 ms serial)
    def step(i, timed):
        st = torch.cuda.current_stream().cuda_stream
        plan.acc.zero_()
        for c in range(nchunks):
            X, Y = bufs[c % 2]
            m = min(chunk, rows_per_gpu - c * chunk)
            _lib.check(L.wm_synth_regression_f32(X.data_ptr(), Y.data_ptr(), m, F, P, 1,
                                                 row_off + c * chunk, bt.data_ptr(), 0.5, 0.01, 3, st))
            if timed:
                e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e[0].record()
            plan.accumulate(X[:m], Y[:m])
            if timed:
                e[1].record()
                pending.append((e, m))
        xch.run(lambda: dp.reduce_gram_(plan.acc), timed)   # the one exchange step: 66x66 float64
        plan.solve()

    total = timed_steps(step, steps, warmup, world, tag="gram64")
    beta = plan.beta.cpu().numpy()
    err = float(np.abs(beta[:F] - beta_true).max() / np.abs(beta_true).max())
    kern_s = sum(a.elapsed_time(b) for (a, b), _ in pending) * 1e-3
    kern_rows = sum(m for _, m in pending)
    krate = kern_rows / kern_s
    bytes_per_row, flops_full = 4 * (F + P), 2 * F * (F + P)
    # what the kernel issues: the 10 upper 16x16 blocks of the symmetric 64x64 Gram on MFMA
    # (2*16*16*10 = 5120 flop per row) + X'y, column sums, y'y on the VALU (2*3*F)
    flops_exec = 2 * 16 * 16 * 10 + 2 * 3 * F
    res = {
        "prewarm_steps": PREWARM_LOG.get("gram64"),
        "value": rows_per_gpu * world * steps / total, "unit": "rows/s",
        "ms_per_step": total / steps * 1e3,
        "kernel_rows_per_s": krate, "kernel_ms_per_chunk": kern_s / len(pending) * 1e3,
        "end_to_end_rows_per_s_per_gpu": rows_per_gpu * steps / total,
        "exchange_us": xch.us(),
        "planted_coef_rel_err": err,
        "config": {"workload": f"multipleRegression least squares F=64, P=1, {rows_per_gpu} rows "
                               f"per GPU generated on the device in {nchunks} resident chunks of "
                               f"{chunk} rows, Gram (fp32 MFMA, fp64 combine) + solve",
                   "rows_per_gpu": rows_per_gpu, "chunk_rows": chunk, "features": F,
                   "full_config_rows_per_gpu": 1_250_000_000,
                   "parallelism": f"dp{world} (row shards by global row index, one 66x66 f64 all-reduce)"},
        "roofline": {"bound": "mfma", "achieved": flops_exec * krate / 1e12,
                     "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops_exec * krate / 1e12 / MFMA_F32_PEAK_TF,
                     "traffic": pmc_traffic(["gram_x64_kernel"], chunk == 50_000_000),
                     "traffic_source": pmc_source(["gram_x64_kernel"], chunk == 50_000_000),
                     "traffic_note": "HBM bytes per 5e7-row chunk launch (algorithmic: 260 B/row)",
                     "kernel": "gram_x64_kernel",
                     "flops_per_row_executed": flops_exec,
                     "flops_per_row_full_product": flops_full,
                     "note": "frac uses the flops the kernel executes (10 of the 16 blocks of the "
                             "symmetric Gram); by SURVEY 8d's full-product count the rate is "
                             f"{flops_full * krate / 1e12:.1f} TFLOP/s",
                     "hbm_GBs": bytes_per_row * krate / 1e9,
                     "hbm_frac": bytes_per_row * krate / 1e9 / HBM_PEAK_GBS}}
    if cpu_base and rank == 0 and world == 1:
        res.update(gram_cpu_baselines(bufs[(nchunks - 1) % 2][0], bufs[(nchunks - 1) % 2][1], F, P,
                                      beta))
    del bufs
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------ convolution
def run_conv(steps, warmup, rank, world, cpu_base, N=64, H=721, W=1440):
    from weather_model_amd import ops, _lib, dp
    L = _lib.lib()
    nset = 3        # rotate buffer sets so the 256 MiB Infinity Cache cannot serve the reads
    w = ops.synth_uniform((1, 1, 3, 3), seed=3, lo=-1 / 3, hi=1 / 3)
    b = ops.synth_uniform((1,), seed=5, lo=-1 / 3, hi=1 / 3)
    xs = [ops.synth_uniform((N, 1, H, W), seed=2, offset=(rank * nset + k) * N * H * W)
          for k in range(nset)]
    gys = [ops.synth_uniform((N, 1, H - 2, W - 2), seed=4, lo=-1.0, hi=1.0,
                             offset=(rank * nset + k) * N * H * W) for k in range(nset)]
    ys = [torch.empty((N, 1, H - 2, W - 2), device="cuda") for _ in range(nset)]
    nbytes = L.wm_conv2d_bwd_wb_workspace_bytes(N, 1, H, W, 1, 3, 3)
    ws = torch.empty((nbytes,), device="cuda", dtype=torch.uint8)
    grads = torch.zeros((10,), device="cuda")
    fwd_ev, bwd_ev = [], []
    pair_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    xch = Exchange(world)

    def launch_pair(k, st, ef=False, eb=False):
        if ef:
            f0, f1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            f0.record()
        _lib.check(L.wm_conv2d_fwd_f32(xs[k].data_ptr(), w.data_ptr(), b.data_ptr(),
                                       ys[k].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))
        if ef:
            f1.record()
            fwd_ev.append((f0, f1))
        if eb:
            b0, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            b0.record()
        _lib.check(L.wm_conv2d_bwd_wb_f32(xs[k].data_ptr(), gys[k].data_ptr(), grads.data_ptr(),
                                          grads[9:].data_ptr(), ws.data_ptr(), ws.numel(),
                                          N, 1, H, W, 1, 3, 3, st))
        if eb:
            b1.record()
            bwd_ev.append((b0, b1))

    def step(i, ev):
        # N = 1: the whole timed region (K x forward + backward, nothing else on the stream) sits
        # between ONE pair of HIP events -- an event record costs ~6 us on the stream, so brackets
        # inside a 190 us step would be 3-6 % of it.  N > 1: the gradient all-reduce sits between
        # the steps, so the kernels are bracketed individually on a quarter of the steps.
        k = i % nset
        st = torch.cuda.current_stream().cuda_stream
        if world == 1:
            if ev and i == warmup:
                pair_ev[0].record()
            launch_pair(k, st)
            if ev and i == warmup + steps - 1:
                pair_ev[1].record()
            return
        mod = 4 if steps >= 8 else (2 if steps >= 2 else 1)
        launch_pair(k, st, ev and i % mod == 0, ev and i % mod == mod // 2)
        xch.run(lambda: dp.reduce_grads_(grads), ev)     # DP weight-gradient exchange (10 floats)

    total = timed_steps(step, steps, warmup, world, tag="conv3x3")
    pts = N * H * W
    out_pts = N * (H - 2) * (W - 2)
    fwd_bytes = 4 * (pts + out_pts)
    bwd_bytes = 4 * (pts + out_pts)
    if world == 1:
        pair_ms = pair_ev[0].elapsed_time(pair_ev[1]) / steps
        # per-kernel figures from a few individually bracketed launches AFTER the timed region
        # (each includes the event overhead; the rocprof summary under profiles/ has the clean ones)
        for i in range(8):
            launch_pair(i % nset, torch.cuda.current_stream().cuda_stream, True, True)
        torch.cuda.synchronize()
    fwd_ms = float(np.median([a.elapsed_time(c) for a, c in fwd_ev]))
    bwd_ms = float(np.median([a.elapsed_time(c) for a, c in bwd_ev]))
    if world > 1:
        pair_ms = fwd_ms + bwd_ms
    default_shape = (N, H, W) == (64, 721, 1440)
    kernels = ["conv_fwd_c1f_kernel<3, 3>", "conv_bwd_wb_c1f_kernel<3, 3>"]
    res = {
        "prewarm_steps": PREWARM_LOG.get("conv3x3"),
        "value": pts * world * steps / total, "unit": "grid-pts/s", "ms_per_step": total / steps * 1e3,
        "config": {"workload": "conv 3x3 C=1 fwd+bwd(weight,bias), 721x1440 grid, batch 64 per GPU"},
        "fwd_ms": fwd_ms, "bwd_wb_ms": bwd_ms, "pair_ms": pair_ms, "exchange_us": xch.us(),
        "roofline": {"bound": "hbm",
                     "achieved": (fwd_bytes + bwd_bytes) / (pair_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "traffic": pmc_traffic(kernels, default_shape),
                     "traffic_source": pmc_source(kernels, default_shape),
                     "kernel": "conv_fwd_c1f_kernel<3,3> + conv_bwd_wb_c1f_kernel<3,3> (the "
                               "weight-gradient finalize launch is inside the backward call and "
                               "its time is counted); "
                               "duration = HIP-event time of the timed region / steps",
                     "fwd_GBs": fwd_bytes / (fwd_ms * 1e-3) / 1e9,
                     "bwd_wb_GBs": bwd_bytes / (bwd_ms * 1e-3) / 1e9,
                     "algorithmic_bytes_per_point": 16},
    }
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS

    if world == 1:
        # deferred finalize (wm_conv2d_bwd_wb_partials_f32 + wm_adam_step_partials_f32): the
        # dependent finalize launch leaves the pair and is absorbed by the optimizer step that
        # follows (newConvolution.py:184-186).  Same bits (tests/test_gpu_conv.py); the headline
        # roofline above stays on the default entry points.
        import ctypes
        prm = torch.cat([w.reshape(-1), b.reshape(-1)]).clone()
        mm, vv = torch.zeros_like(prm), torch.zeros_like(prm)
        layout = (ctypes.c_int * 2)()
        lay_p = ctypes.cast(layout, ctypes.c_void_p)

        def fwd_(k, st):
            _lib.check(L.wm_conv2d_fwd_f32(xs[k].data_ptr(), w.data_ptr(), b.data_ptr(),
                                           ys[k].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))

        def partials_(k, st):
            _lib.check(L.wm_conv2d_bwd_wb_partials_f32(
                xs[k].data_ptr(), gys[k].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3,
                lay_p, st))

        def default_step(i, st):
            launch_pair(i % nset, st)
            _lib.check(L.wm_adam_step_f32(prm.data_ptr(), grads.data_ptr(), mm.data_ptr(),
                                          vv.data_ptr(), 10, i + 1, 1e-3, 0.9, 0.999, 1e-8, 1.0, st))

        def deferred_pair(i, st):
            fwd_(i % nset, st)
            partials_(i % nset, st)

        def deferred_step(i, st):
            deferred_pair(i, st)
            _lib.check(L.wm_adam_step_partials_f32(
                prm.data_ptr(), ws.data_ptr(), layout[0], layout[1], mm.data_ptr(), vv.data_ptr(),
                10, grads.data_ptr(), i + 1, 1e-3, 0.9, 0.999, 1e-8, 1.0, st))

        def region(fn):
            st = torch.cuda.current_stream().cuda_stream
            e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            for i in range(3):
                fn(i, st)
            e0.record()
            for i in range(steps):
                fn(i, st)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps

        dpair, sdef, sdfr = region(deferred_pair), region(default_step), region(deferred_step)
        res["deferred_finalize"] = {
            "pair_ms": dpair, "frac": (fwd_bytes + bwd_bytes) / (dpair * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_default_ms": sdef, "step_deferred_ms": sdfr,
            "note": "pair = forward + partial-sum pass (finalize deferred); step = pair + Adam on "
                    "the 10 parameters: default = wm_conv2d_bwd_wb_f32 + wm_adam_step_f32 (4 "
                    "launches), deferred = ..._partials_f32 + wm_adam_step_partials_f32 (3 launches)"}

    # the 24 B/pt variant of SURVEY 8d row 3: the input needs a gradient too (backward-data)
    gxs = [torch.empty_like(xs[k]) for k in range(nset)]
    def step3(i, ev):
        k = i % nset
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(L.wm_conv2d_fwd_f32(xs[k].data_ptr(), w.data_ptr(), b.data_ptr(),
                                       ys[k].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))
        _lib.check(L.wm_conv2d_bwd_wb_f32(xs[k].data_ptr(), gys[k].data_ptr(), grads.data_ptr(),
                                          grads[9:].data_ptr(), ws.data_ptr(), ws.numel(),
                                          N, 1, H, W, 1, 3, 3, st))
        _lib.check(L.wm_conv2d_bwd_data_f32(gys[k].data_ptr(), w.data_ptr(), gxs[k].data_ptr(),
                                            N, 1, H, W, 1, 3, 3, st))
        if world > 1:
            dp.reduce_grads_(grads)

    w3 = max(1, warmup // 2)
    trio_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def step3b(i, ev):
        if ev and i == w3:
            trio_ev[0].record()
        step3(i, ev)
        if ev and i == w3 + steps - 1:
            trio_ev[1].record()

    total3 = timed_steps(step3b, steps, w3, world, tag="trio")
    trio_ms = trio_ev[0].elapsed_time(trio_ev[1]) / steps
    bd_ms = trio_ms - pair_ms
    res["with_bwd_data"] = {
        "value": pts * world * steps / total3, "unit": "grid-pts/s",
        "ms_per_step": total3 / steps * 1e3, "trio_ms": trio_ms,
        "bwd_data_ms": bd_ms,   # trio - pair (at N > 1 the trio bracket also holds the all-reduce)
        "bwd_data_GBs": fwd_bytes / (bd_ms * 1e-3) / 1e9,
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "achieved": 3 * fwd_bytes / (trio_ms * 1e-3) / 1e9,
                     "frac": 3 * fwd_bytes / (trio_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(kernels + ["conv_bwd_data_c1f_kernel<3, 3>"],
                                            default_shape),
                     "traffic_source": pmc_source(kernels + ["conv_bwd_data_c1f_kernel<3, 3>"],
                                                  default_shape),
                     "kernel": "+ conv_bwd_data_c1f_kernel<3,3>",
                     "algorithmic_bytes_per_point": 24}}
    del gxs
    if cpu_base and rank == 0 and world == 1:
        Nc = 8
        nthr = cpu_threads()
        torch.set_num_threads(nthr)
        xc = xs[0][:Nc].cpu()
        gyc = gys[0][:Nc].cpu()
        wc = w.cpu().requires_grad_(True)
        bc = b.cpu().requires_grad_(True)

        def lib():
            yc = torch.nn.functional.conv2d(xc, wc, bc)
            torch.autograd.grad(yc, (wc, bc), gyc)

        med, reps = median_time(lib, 8.0, min_reps=3, max_reps=30)
        res["cpu_baseline"] = {
            "value": Nc * H * W / med, "unit": "grid-pts/s", "cores": nthr, "kind": "library",
            "cpu": cpu_model_name(),
            "sample": f"{Nc} of the 64 images: torch.nn.functional.conv2d + torch.autograd.grad("
                      f"weight, bias) on CPU tensors, {nthr} threads, median of {reps}"}
        from oracle import cpu as ocpu
        xh, gyh = xc.numpy(), gyc.numpy()
        wh, bh = w.cpu().numpy(), b.cpu().numpy()
        medp, repsp = median_time(lambda: ocpu.conv2d_fwd_bwd_wb_f32(xh, wh, bh, gyh), 4.0, 2, 30)
        res["cpu_baseline_port"] = {
            "value": Nc * H * W / medp, "unit": "grid-pts/s", "cores": ocpu.threads(),
            "kind": "port",
            "sample": f"{Nc} of the 64 images, oracle/wm_oracle.c float32 forward + float64 "
                      f"backward(weight,bias), OpenMP, median of {repsp}"}
    return res


# ------------------------------------------------------------------------------ fused model
def run_model(steps, warmup, rank, world, cpu_base, B=721 * 1440):
    """BASELINE configs[3] (4a): the reference-shaped 32-channel conv model, forward + backward +
    Adam on one 721x1440 slice of grid points per GPU per step; DP = one gradient all-reduce."""
    from weather_model_amd import ops, synth, _lib, dp
    L = _lib.lib()
    parts = []
    for k, (name, shp) in enumerate(ops.MODEL_PARAM_SHAPES):
        fan = int(np.prod(shp[1:])) if len(shp) > 1 else {16: 2, 32: 48 if "conv2" in name else 96,
                                                          64: 192, 6: 64}[shp[0]]
        bnd = 1.0 / np.sqrt(fan)
        parts.append(synth.uniform(900 + k, shp, -bnd, bnd).reshape(-1))
    params = torch.from_numpy(np.concatenate(parts)).cuda()
    m = torch.zeros_like(params)
    v = torch.zeros_like(params)
    x = ops.synth_uniform((B, 2, 6), seed=20, offset=rank * B * 12)
    y = ops.synth_uniform((B, 6), seed=21, offset=rank * B * 6)
    out = torch.empty((B, 6), device="cuda")
    grads = torch.empty_like(params)
    loss = torch.zeros((), device="cuda")
    ws = torch.empty((L.wm_model_workspace_bytes(B),), device="cuda", dtype=torch.uint8)
    evs = []
    state = {"step": 0}

    trace = bool(os.environ.get("WM_BENCH_TRACE"))

    xch = Exchange(world)

    def step(i, timed):
        st = torch.cuda.current_stream().cuda_stream
        if trace:
            torch.cuda.synchronize()
            print(f"[rank {rank}] model step {i} start", file=sys.stderr, flush=True)
        if timed:
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e[0].record()
        _lib.check(L.wm_model_train_f32(params.data_ptr(), x.data_ptr(), y.data_ptr(), 0,
                                        out.data_ptr(), loss.data_ptr(), grads.data_ptr(),
                                        ws.data_ptr(), ws.numel(), B, B * world, st))
        if timed:
            e[1].record()
            evs.append(e)
        if trace:
            torch.cuda.synchronize()
            print(f"[rank {rank}] model step {i} kernel done", file=sys.stderr, flush=True)
        xch.run(lambda: dp.reduce_grads_(grads), timed)   # 82 KB flat gradient buffer, one collective
        if trace:
            torch.cuda.synchronize()
            print(f"[rank {rank}] model step {i} reduced", file=sys.stderr, flush=True)
        state["step"] += 1
        _lib.check(L.wm_adam_step_f32(params.data_ptr(), grads.data_ptr(), m.data_ptr(),
                                      v.data_ptr(), params.numel(), state["step"], 1e-3, 0.9,
                                      0.999, 1e-8, 1.0, st))

    loss0 = None
    total = timed_steps(step, steps, warmup, world, tag="model32")
    kms = float(np.mean([a.elapsed_time(c) for a, c in evs]))
    flops = 353664.0                       # algorithmic fwd+bwd flops per sample
    ach = flops * B / (kms * 1e-3) / 1e12
    res = {"prewarm_steps": PREWARM_LOG.get("model32"), "value": B * world * steps / total, "unit": "grid-pts/s",
           "ms_per_step": total / steps * 1e3, "train_kernel_ms": kms,
           "final_loss": float(loss) * world, "exchange_us": xch.us(),
           "config": {"workload": "newConvolution.CustomModel (1->16->32->32->32 conv stack + "
                                  "192->64->6 head) fwd+bwd+Adam, 1038240 grid points per GPU per step"},
           "roofline": {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF,
                        "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF,
                        "traffic": pmc_traffic(["model_kernel<1, false>"], B == 721 * 1440),
                        "traffic_source": pmc_source(["model_kernel<1, false>"], B == 721 * 1440),
                        "kernel": "model_kernel<1>", "algorithmic_flops_per_sample": flops,
                        # the kernel issues 101.6 M v_mfma_f32_16x16x4_f32 (2048 flop each) per
                        # 1 038 240-sample step: the start/end rows of every conv input are rank-1
                        # terms, not contraction rows -- 57 % of the algorithmic count
                        "executed_flops_per_sample": EXEC_FLOPS_PER_SAMPLE,
                        "executed_TFLOPs": EXEC_FLOPS_PER_SAMPLE * B / (kms * 1e-3) / 1e12,
                        "executed_frac_of_peak": EXEC_FLOPS_PER_SAMPLE * B / (kms * 1e-3) / 1e12
                        / MFMA_F32_PEAK_TF}}
    # the same step fed from a resident (T,1,721,1440,6) volume by sample index: no gather launch,
    # no x / y arrays (wm_model_train_gather_f32)
    if B == 721 * 1440:
        T = 9
        vol = ops.synth_uniform((T, 1, 721, 1440, 6), seed=22, offset=rank * T * B * 6)
        gev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        nrep = max(3, min(steps, 10))
        for i in range(nrep + 2):
            if i == 2:
                gev[0].record()
            _lib.check(L.wm_model_train_gather_f32(
                params.data_ptr(), vol.data_ptr(), 0, 0, T, 1, 721, 1440, T - 6, 1, 721, 1440, 0, 3,
                6, out.data_ptr(), loss.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(), B,
                B * world, torch.cuda.current_stream().cuda_stream))
        gev[1].record()
        torch.cuda.synchronize()
        gms = gev[0].elapsed_time(gev[1]) / nrep
        res["gather_fed"] = {"train_ms": gms, "value_per_gpu": B / (gms * 1e-3),
                             "note": "forward + backward of the same batch read from the volume by "
                                     "index (replaces wm_gather_triplets_f32 + wm_model_train_f32)"}
        del vol
    if cpu_base and rank == 0 and world == 1:
        pd, off = {}, 0
        pc = params.cpu().numpy()
        for name, shp in ops.MODEL_PARAM_SHAPES:
            k = int(np.prod(shp))
            pd[name] = pc[off:off + k].reshape(shp)
            off += k
        # the library path the reference runs (newConvolution.py:174-186): eager torch on the CPU,
        # one grid point per step -- forward, MSE, backward, torch.optim.Adam, zero_grad
        from oracle import model_torch
        nthr = cpu_threads()
        torch.set_num_threads(nthr)
        em = model_torch.EagerModel(pd)
        opt = torch.optim.Adam(em.parameters())
        ns = 400
        xc, yc = x[:ns].cpu().unsqueeze(1), y[:ns].cpu().unsqueeze(1)
        model_torch.train_steps(em, xc[:20], yc[:20], opt)            # warm-up
        per = []
        t_end = time.perf_counter() + 8.0
        while len(per) < 3 or (time.perf_counter() < t_end and len(per) < 20):
            t0 = time.perf_counter()
            model_torch.train_steps(em, xc, yc, opt)
            per.append((time.perf_counter() - t0) / ns)
        res["cpu_baseline"] = {
            "value": 1.0 / float(np.median(per)), "unit": "grid-pts/s", "cores": nthr,
            "kind": "library", "cpu": cpu_model_name(),
            "sample": f"{ns} grid points per pass, batch 1: oracle/model_torch.py (eager torch-CPU "
                      f"restatement of CustomModel + MSELoss + backward + torch.optim.Adam per "
                      f"sample, as newConvolution.py:174-186), median of {len(per)} passes"}
        from oracle import model_ref
        nb = 16384
        xs, ys = x[:nb].cpu().numpy(), y[:nb].cpu().numpy()
        medp, repsp = median_time(lambda: model_ref.loss_and_grads(pd, xs, ys), 4.0, 2, 30)
        res["cpu_baseline_port"] = {
            "value": nb / medp, "unit": "grid-pts/s", "cores": nthr, "kind": "port",
            "sample": f"{nb} samples, oracle/model_ref.py (numpy float64 BATCHED fwd+bwd, BLAS "
                      f"threads), median of {repsp}"}
    return res


# ------------------------------------------------------------------------------ 32-channel conv
def conv_stack_cpu_library(ws, bs, x1, g1, relu, reps=5):
    """Library CPU baseline of the 32-channel layers: torch.nn.functional.conv2d (+ ReLU) forward and
    torch.autograd.grad of the weights, biases and the input on ONE image of the GPU run's data."""
    nthr = cpu_threads()
    torch.set_num_threads(nthr)
    ws = [t.clone().requires_grad_(True) for t in ws]
    bs = [t.clone().requires_grad_(True) for t in bs]
    xin = x1.clone().requires_grad_(True)

    def one():
        h = xin
        for wl, bl in zip(ws, bs):
            h = torch.nn.functional.conv2d(h, wl, bl)
            if relu:
                h = torch.relu(h)
        torch.autograd.grad(h, [xin] + ws + bs, g1)

    one()
    t, n = median_time(one, 8.0, min_reps=3, max_reps=reps)
    pts = x1.shape[0] * x1.shape[2] * x1.shape[3]
    return {"value": pts / t, "unit": "grid-pts/s", "cores": nthr, "kind": "library",
            "cpu": cpu_model_name(),
            "sample": f"1 of the images per GPU: torch.nn.functional.conv2d"
                      f"{' + relu' if relu else ''} x{len(ws)} forward + torch.autograd.grad(input, "
                      f"weights, biases) on CPU tensors, {nthr} threads, median of {n}"}


CONV32_KERNELS = ["conv3d_fwd_rows_kernel<2>", "conv_ring_kernel<3, 3, 8, true>",
                  "conv3d_bwd_wb_staged_kernel<1, 9, 2>"]
CONV32_PMC = os.path.join("profiles", "r03", "pmc_traffic_conv32.json")


def run_conv32(steps, warmup, rank, world, cpu_base, N=8, C=32, H=721, W=1440):
    """BASELINE configs[3] read literally (4b): one Conv(32->32, 3x3) layer on (N,32,721,1440),
    forward + backward(data, weight, bias) -- the implicit-GEMM MFMA kernels."""
    from weather_model_amd import ops, dp, _lib
    x = ops.synth_uniform((N, C, H, W), seed=30, offset=rank * N * C * H * W)
    w = ops.synth_uniform((C, C, 3, 3), seed=31, lo=-1.0 / 17, hi=1.0 / 17)
    b = ops.synth_uniform((C,), seed=32, lo=-1.0 / 17, hi=1.0 / 17)
    gy = ops.synth_uniform((N, C, H - 2, W - 2), seed=33, lo=-1.0, hi=1.0,
                           offset=rank * N * C * (H - 2) * (W - 2))
    flat = torch.zeros((C * C * 9 + C,), device="cuda")
    evs = {"fwd": [], "bwd_data": [], "bwd_wb": []}

    xch = Exchange(world)

    def step(i, timed):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            marks[0].record()
        ops.conv2d_forward(x, w, b)
        if timed:
            marks[1].record()
        ops.conv2d_backward_data(gy, w, x.shape)
        if timed:
            marks[2].record()
        gw, gb = ops.conv2d_backward_wb(x, gy, w.shape)
        if timed:
            marks[3].record()
            for k, (a, c) in zip(evs, zip(marks[:-1], marks[1:])):
                evs[k].append((a, c))
        if world > 1:
            flat[:C * C * 9] = gw.reshape(-1)
            flat[C * C * 9:] = gb
            xch.run(lambda: dp.reduce_grads_(flat), timed)

    total = timed_steps(step, steps, warmup, world, tag="conv32")
    ms = {k: float(np.mean([a.elapsed_time(c) for a, c in v])) for k, v in evs.items()}
    # the clock the chip held under the weight-gradient kernel's MFMA load (the kernel's first
    # workgroup leaves its cycle and wall-clock counts behind the partial sums)
    clock_mhz = None
    wsb = ops.conv2d_bwd_wb_workspace(x.shape, w.shape)
    off = _lib.lib().wm_conv3d_bwd_wb_clock_offset(N, C, 1, H, W, C, 1, 3, 3)
    if off:
        for _ in range(3):
            ops.conv2d_backward_wb(x, gy, w.shape, ws=wsb)
        rec = wsb[off:off + 24].cpu().numpy().view(np.uint64)
        if int(rec[0]) == 0x574d434c4f434b31 and rec[2] > 0:
            clock_mhz = float(rec[1]) / float(rec[2]) * 100.0
    pts = N * H * W
    flops_pass = 2.0 * C * C * 9 * N * (H - 2) * (W - 2)
    ach = 3 * flops_pass / (sum(ms.values()) * 1e-3) / 1e12
    base = None
    if cpu_base and rank == 0 and world == 1:
        base = conv_stack_cpu_library([w.cpu()], [b.cpu()], x[:1].cpu(), gy[:1].cpu(), relu=False)
    return {"prewarm_steps": PREWARM_LOG.get("conv32"), **({"cpu_baseline": base} if base else {}),
            "exchange_us": xch.us(),
            "value": pts * world * steps / total, "unit": "grid-pts/s",
            "ms_per_step": total / steps * 1e3, "kernel_ms": ms,
            "config": {"workload": f"Conv2d({C}->{C}, 3x3) fwd + bwd(data, weight, bias) on "
                                   f"({N},{C},{H},{W}) per GPU"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF,
                         "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF,
                         # (the PMC passes of THIS workload alone: in the full line's passes the ring
                         # kernel's average also holds the masked calls of conv32x3)
                         "traffic": pmc_traffic(CONV32_KERNELS, (N, C, H, W) == (8, 32, 721, 1440),
                                                CONV32_PMC),
                         "traffic_source": pmc_source(CONV32_KERNELS,
                                                      (N, C, H, W) == (8, 32, 721, 1440), CONV32_PMC),
                         "algorithmic_bytes": 4.0 * N * C * (3 * H * W + 3 * (H - 2) * (W - 2)),
                         "kernel": "conv3d_fwd_rows_kernel<2> + conv_ring_kernel<3,3,8,true> + "
                                   "conv3d_bwd_wb_staged_kernel",
                         "algorithmic_flops_per_point_per_pass": 2 * C * C * 9,
                         "TFLOPs": {k: flops_pass / (v * 1e-3) / 1e12 for k, v in ms.items()},
                         "shader_clock_mhz_under_bwd_wb": clock_mhz}}


def run_conv32x3(steps, warmup, rank, world, cpu_base, N=8, C=32, H=721, W=1440, L=3):
    """BASELINE configs[3] read literally, as a MODEL: L = 3 layers of Conv2d(32 -> 32, 3x3) + ReLU
    on (N, 32, 721, 1440) per GPU, forward + backward (weight, bias gradients of every layer,
    input gradients between the layers with the ReLU mask fused into the backward-data store), one
    flat gradient all-reduce of 3 x 9248 floats under data parallelism."""
    from weather_model_amd import ops
    from weather_model_amd.stack import ConvStack
    x = ops.synth_uniform((N, C, H, W), seed=40, offset=rank * N * C * H * W)
    ws = [ops.synth_uniform((C, C, 3, 3), seed=41 + l, lo=-1.0 / 17, hi=1.0 / 17) for l in range(L)]
    bs = [ops.synth_uniform((C,), seed=45 + l, lo=-0.05, hi=0.05) for l in range(L)]
    st = ConvStack(ws, bs, x.shape)
    oshape = st.shapes[-1]
    gout = ops.synth_uniform(oshape, seed=49, lo=-1.0, hi=1.0,
                             offset=rank * int(np.prod(oshape)))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def step(i, timed):
        if timed and i == warmup:
            ev[0].record()
        st.step_grads(x, gout)             # gout is read only: its masked copy is the stack's
        if timed and i == warmup + steps - 1:
            ev[1].record()

    total = timed_steps(step, steps, warmup, world, tag="conv32x3")
    ms = ev[0].elapsed_time(ev[1]) / steps
    # 2 C C 9 flop per output position: L forward passes, L weight-gradient passes over the same
    # positions, L - 1 backward-data passes (over the INPUT positions of layers 2..L)
    flops = 0.0
    for l in range(1, L + 1):
        out_pos = float(np.prod(st.shapes[l])) / C
        flops += 2.0 * C * C * 9 * out_pos * 2
        if l > 1:
            flops += 2.0 * C * C * 9 * out_pos          # executed over the outputs of layer l
    ach = flops / (ms * 1e-3) / 1e12
    clipped = float((st.h[L] == 0).float().mean())
    base = None
    if cpu_base and rank == 0 and world == 1:
        base = conv_stack_cpu_library([t.cpu() for t in ws], [t.cpu() for t in bs], x[:1].cpu(),
                                      gout[:1].cpu(), relu=True, reps=3)
    return {"prewarm_steps": PREWARM_LOG.get("conv32x3"), **({"cpu_baseline": base} if base else {}),
            "value": N * H * W * world * steps / total, "unit": "grid-pts/s",
            "ms_per_step": total / steps * 1e3, "step_kernels_ms": ms,
            "relu_clipped_fraction_last_layer": clipped,
            "grad_floats_all_reduced": int(st.flat_grads.numel()),
            "config": {"workload": f"{L} x [Conv2d({C}->{C}, 3x3) + ReLU] forward + backward on "
                                   f"({N},{C},{H},{W}) per GPU, ReLU masks fused, one flat gradient "
                                   f"all-reduce"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF,
                         "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": None,
                         "kernel": "conv3d_fwd_rows_kernel x3 + conv3d_bwd_wb_staged_kernel x3 + "
                                   "conv3d_bwd_data_rows_kernel x2 (+ mask, copy, finalize launches "
                                   "inside the bracket)",
                         "algorithmic_flops_per_step": flops}}


# ------------------------------------------------------------------------------ Conv3D model
def run_conv3d_model(steps, warmup, batch=16, world=1):
    """the reference's Conv3D network (convolution.py:156-180) at its own shapes,
    input (1,20,20,20), batch 16: one training step (two conv branches forward and backward, pools,
    two dense layers, Adam) replayed from a HIP graph.  The reference path has no data-parallel
    exchange: at N > 1 every rank trains its own replica (replicas only, no collective) and the
    value is the sum over ranks by the slowest rank's clock."""
    from weather_model_amd.dropin._conv3d_model import Conv3DModel
    m = Conv3DModel((1, 20, 20, 20), seed=0)
    m.compile()
    gen = torch.Generator(device="cuda").manual_seed(7)
    x1 = torch.rand((batch, 1, 20, 20, 20), device="cuda", generator=gen)
    x2 = torch.rand((batch, 1, 20, 20, 20), device="cuda", generator=gen)
    y = torch.rand((batch, 1, 8000), device="cuda", generator=gen)
    for _ in range(max(warmup, 3)):
        m.graphed_train_step((x1, x2, y))
    torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    pending = None
    for _ in range(steps):                      # as Conv3DModel.fit drives it: logs read one step late
        issued = m.graphed_train_step((x1, x2, y), defer=True)
        if pending is not None:
            logs = pending.result()
        pending = issued
    logs = pending.result()
    torch.cuda.synchronize()
    dt = max_over_ranks(time.perf_counter() - t0, world) / steps
    # 2 branches x (layer 1: 16*14^3 outputs x 32 x 343, layer 2: 16*3*3*10 x 64 x 2000) MACs forward,
    # the same again for the weight gradients, layer 2 once more for its input gradient; dense
    # layers 640x640 and 640x8000 forward + 2x backward
    conv = 2 * 2.0 * batch * (14 ** 3 * 32 * 343 + 90 * 64 * 2000)
    flops = 2 * conv + 2 * 2.0 * batch * 90 * 64 * 2000 + 3 * 2.0 * batch * (640 * 640 + 640 * 8000)
    return {"value": batch * world / dt, "unit": "samples/s", "ms_per_step": dt * 1e3,
            "loss": logs["loss"], "exchange_us": None,
            "config": {"workload": f"Conv3D interpolation model, input (1,20,20,20), batch {batch}, "
                                   "fit step (HIP-graph replay, every step's loss and metrics "
                                   "read back, one step late as in fit())",
                       "parallelism": f"{world} independent replica(s), no collective"},
            "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_F32_PEAK_TF,
                         "unit": "TFLOP/s", "frac": flops / dt / 1e12 / MFMA_F32_PEAK_TF,
                         "traffic": None, "kernel": "conv3d split-K / small-volume kernels (gather-"
                                                    "and launch-bound at these sizes)",
                         "algorithmic_flops_per_step": flops}}


# ------------------------------------------------------------------------------ field kernels
def run_fields(steps, warmup, rank, world, cpu_base):
    """one-pass statistics, normalise and the sample gather, each on a
    64 x 721 x 1440 field (266 MB, 3 rotating copies) / one 721 x 1440 slice of samples; row a9:
    the regression generators' rows from a resident volume."""
    from weather_model_amd import ops
    n = 64 * 721 * 1440
    nset = 3
    fields = [ops.synth_uniform((n,), seed=70 + k, lo=-50.0, hi=1300.0, offset=rank * n)
              for k in range(nset)]
    outs = [torch.empty_like(f) for f in fields]
    T, Z, Y, X, V = 9, 1, 721, 1440, 6
    vol = ops.synth_uniform((T, Z, Y, X, V), seed=80, offset=rank * T * Z * Y * X * V)
    B = Y * X
    stat = torch.zeros((5,), device="cuda", dtype=torch.float64)

    def timed(fn):
        ev = []

        def step(i, t):
            if t:
                e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e[0].record()
            fn(i)
            if t:
                e[1].record()
                ev.append(e)
        total = timed_steps(step, steps, warmup, world, tag="fields")
        return total / steps, float(np.mean([a.elapsed_time(c) for a, c in ev])) * 1e-3

    t_stat, k_stat = timed(lambda i: ops.field_stats(fields[i % nset], stat))
    t_norm, k_norm = timed(lambda i: ops.affine_scale(fields[i % nset], -50.0, 1350.0,
                                                      out=outs[i % nset]))
    t_gath, k_gath = timed(lambda i: ops.gather_triplets(vol, (T - 6, Z, Y, X), first=0, count=B))
    res = {
        "prewarm_steps": PREWARM_LOG.get("fields"),
        "unit": "elements/s",
        "config": {"workload": "field kernels: stats + normalise over 64x721x1440 fp32, triplet "
                               "gather of one 721x1440 slice (6 variables), regression rows from a "
                               "resident volume"},
        "stats": {"value": n * world / t_stat, "ms": t_stat * 1e3,
                  "roofline": {"bound": "hbm", "achieved": 4 * n / k_stat / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": 4 * n / k_stat / 1e9 / HBM_PEAK_GBS,
                               "traffic": pmc_traffic(["field_stats_kernel"], True),
                               "traffic_source": pmc_source(["field_stats_kernel"], True),
                               "kernel": "field_stats_kernel", "algorithmic_bytes_per_element": 4}},
        "normalise": {"value": n * world / t_norm, "ms": t_norm * 1e3,
                      "roofline": {"bound": "hbm", "achieved": 8 * n / k_norm / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": 8 * n / k_norm / 1e9 / HBM_PEAK_GBS,
                                   "traffic": pmc_traffic(["affine_scale_kernel"], True),
                                   "traffic_source": pmc_source(["affine_scale_kernel"], True),
                                   "kernel": "affine_scale_kernel",
                                   "algorithmic_bytes_per_element": 8}},
        "gather": {"value": B * world / t_gath, "unit": "samples/s", "ms": t_gath * 1e3,
                   "roofline": {"bound": "hbm", "achieved": 144 * B / k_gath / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": 144 * B / k_gath / 1e9 / HBM_PEAK_GBS,
                                "traffic": pmc_traffic(["gather_triplets_u32_kernel<2>"], True),
                                "traffic_source": pmc_source(["gather_triplets_u32_kernel<2>"], True),
                                "kernel": "gather_triplets_u32_kernel<2>",
                                "algorithmic_bytes_per_sample": 144}},
    }
    # row a9: design-matrix rows of the regression generators (cube 1, 8 variables, 16 rows per
    # batch, the region walk in closed form) from a resident volume stored point-major
    del vol
    files, nv, Zc, bs, length, count = 70, 8, 2, 16, 4, 65536
    stored = ops.synth_uniform((Zc, Y, X, files, nv), seed=81, offset=rank * Zc * Y * X * files * nv)
    pm = stored.permute(3, 4, 0, 1, 2)                  # logical (files, vars, Z, Y, X)
    walk = {"batch": bs, "length": length, "index": 0, "region": (0, 0, 0), "dims": (Zc, Y, X)}
    crow = count * bs
    t_cub, k_cub = timed(lambda i: ops.gather_cubes(pm, (1, 1, 1), 3, 6, crow, walk=walk))
    res["cubes"] = {"value": crow * world / t_cub, "unit": "rows/s", "ms": t_cub * 1e3,
                    "roofline": {"bound": "hbm", "achieved": 24 * nv * crow / k_cub / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": 24 * nv * crow / k_cub / 1e9 / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic(["gather_cubes_kernel<true, 4>"], True), "traffic_source": pmc_source(["gather_cubes_kernel<true, 4>"], True), "kernel": "gather_cubes_kernel<true, 4>",
                                 "algorithmic_bytes_per_row": 24 * nv},
                    "config": {"workload": f"regression rows [x(t) | x(t+6) ; x(t+3)]: {crow} rows "
                                           f"of {nv} variables, cube 1, batches of {bs}, region walk "
                                           "of multipleRegression.py:259-270 in closed form",
                               "volume_layout": "point-major (Z,Y,X,files,vars), passed as a strided "
                                                "view; the reference's row-major order makes every "
                                                "element a 4-byte gather (same rows: 289 us vs 40)"}}
    del stored, pm
    res["value"] = res["stats"]["value"]
    res["ms_per_step"] = (t_stat + t_norm + t_gath) * 1e3
    if cpu_base and rank == 0 and world == 1:
        from oracle import cpu as ocpu
        m = 8_000_000
        xs = fields[0][:m].cpu().numpy()
        best = 1e30
        t_end = time.perf_counter() + 5.0
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            ocpu.field_stats(xs)
            ocpu.affine_scale(xs, -50.0, 1350.0)
            best = min(best, time.perf_counter() - t0)
        res["cpu_baseline"] = {"value": m / best, "unit": "elements/s (stats + normalise)",
                               "cores": 1, "kind": "port",
                               "sample": f"{m} elements, oracle/cpu.py numpy restatement"}
    return res


def run_plumbing(steps, warmup, rank, world, n_rows=1_000_000):
    """BASELINE configs[0]: the 1-predictor fit of SURVEY 8d row 1 on the CPU numpy path (no GPU,
    no libwm.so): every rank forms the float64 sums of its row shard, one gloo all-reduce, closed-form
    slope/intercept.  It exercises the launcher, the row sharding and the exchange step only."""
    rng = np.random.default_rng(1000 + rank)
    x = rng.random(n_rows)
    y = 0.7 * x + 0.2 + 0.01 * rng.standard_normal(n_rows)
    beta = [None]

    def step(i, timed):
        acc = torch.tensor([n_rows, x.sum(), y.sum(), (x * x).sum(), (x * y).sum()],
                           dtype=torch.float64)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(acc)
        n, sx, sy, sxx, sxy = acc.tolist()
        slope = (sxy - sx * sy / n) / (sxx - sx * sx / n)
        beta[0] = (slope, (sy - slope * sx) / n)

    total = timed_steps(step, steps, warmup, world, cuda=False)
    return {"value": n_rows * world * steps / total, "ms_per_step": total / steps * 1e3,
            "slope": beta[0][0], "intercept": beta[0][1],
            "planted_coef_max_err": max(abs(beta[0][0] - 0.7), abs(beta[0][1] - 0.2))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="all", choices=["all", "gram16", "conv3x3", "gram64", "model32", "fields", "conv32", "conv32x3",
                             "conv3d_model", "plumbing", "gram_ref_multi", "gram_ref_single", "gram_mid", "gram_family",
                             "gram16_normal", "fit_walk"])
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (gram16)")
    ap.add_argument("--rows64", type=int, default=1_250_000_000,
                    help="rows per GPU of the F=64 workload (config 5: 1e10 rows / 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    if args.workload == "plumbing":
        rank, world, _, dinfo = dist_setup(args.gpus, cpu_only=True)
        r = run_plumbing(args.steps, args.warmup, rank, world)
        if rank == 0:
            print(json.dumps({
                "metric": "regression rows/s (1-predictor fit, CPU numpy plumbing)", "unit": "rows/s",
                "n_gpus": 0, "n_ranks": world, "steps": args.steps, "warmup": args.warmup,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic", "rccl_ranks": dinfo["rccl_ranks"], "backend": dinfo["backend"],
                "launcher": dinfo["launcher"],
                "config": {"workload": "singleRegression 1-predictor least squares, 1e6 rows per "
                                       "rank, numpy float64 on the CPU (BASELINE configs[0])"}, **r}))
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (weather_model_amd has no CPU fallback)")
    rank, world, _, dinfo = dist_setup(args.gpus)
    cpu_base = not args.no_cpu_baseline

    out = {
        "prewarm": f"setup before the W warm-ups of every workload: its own step run untimed for "
                   f"~{PREWARM_MS:.0f} ms of GPU time so the chip is at its steady clocks "
                   f"(prewarm_steps per workload; WM_BENCH_PREWARM_MS=0 turns it off)",
        "metric": "regression rows/s (16-feature least squares, Gram build + solve)",
        "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "rccl_ranks": dinfo["rccl_ranks"], "backend": dinfo["backend"],
        "launcher": dinfo["launcher"],
        "parity": "conv2d / fused model / fields pinned by fixtures from the imported reference; "
                  "regression and Conv3D parity unpinned (keras/tensorflow absent, oracle = "
                  "closed form vs numpy lstsq / torch-CPU)",
    }
    if "rehearsal" in dinfo:
        out["rehearsal"] = dinfo["rehearsal"]
    if args.workload in ("all", "gram16"):
        F, rows = 16, args.rows
        r = run_gram(F, rows, args.steps, args.warmup, rank, world, cpu_base)
        out.update({"value": r["value"], "ms_per_step": r["ms_per_step"],
                    "roofline": r["roofline"],
                    "config": {"workload": f"multipleRegression least squares: F={F}, P=1, "
                                           f"{rows} rows per GPU, X~U(0,1), Gram (fp32 MFMA, fp64 "
                                           f"combine) + normal-equation solve",
                               "rows_per_gpu": rows, "features": F,
                               "parallelism": f"dp{world} (row shards, one {F + 2}x{F + 2} f64 all-reduce)"},
                    "planted_coef_max_err": r["planted_coef_max_err"],
                    "kernel_ms_avg": r["kernel_ms_avg"], "exchange_us": r["exchange_us"],
                    "prewarm_steps": r.get("prewarm_steps")})
        for k in ("cpu_baseline", "cpu_baseline_port", "solve_ms_avg"):
            if k in r:
                out[k] = r[k]
    # the reference's OWN regression shapes, and config 2's second distribution
    for name, (F_, P_, rows_, dist_, what) in {
            "gram_ref_multi": (2, 1, 400_000_000, "uniform",
                               "multipleRegression's own fit (CUBE_SIZE 1, one variable: "
                               "multipleRegression.py:35,273-286)"),
            "gram_ref_single": (432, 216, 1_000_000, "uniform",
                                "singleRegression's own fit (6^3 cubes of T: "
                                "singleRegression.py:21,33,74-85)"),
            "gram_mid": (16, 8, 100_000_000, "uniform",
                         "multipleRegression's fit with eight variables at CUBE_SIZE 1 (F = 2p, P = p, "
                         "p = 8: multipleRegression.py:273-286)"),
            "gram16_normal": (16, 1, args.rows, "normal",
                              "multipleRegression least squares on standardised features "
                              "(multipleRegression.py:195-197)")}.items():
        if args.workload not in ("all", name):
            continue
        kk = args.steps if args.workload == name else min(args.steps, 10)
        ww = args.warmup if args.workload == name else min(args.warmup, 2)
        r = run_gram(F_, rows_, kk, ww, rank, world, cpu_base, P=P_, dist=dist_)
        r["steps"], r["warmup"] = kk, ww
        r["config"] = gram_config(F_, P_, rows_, world, dist_, what)
        if args.workload == name:
            out.update({"metric": f"regression rows/s ({what})", "value": r["value"],
                        "ms_per_step": r["ms_per_step"], "roofline": r["roofline"],
                        "config": r["config"], "kernel_ms_avg": r["kernel_ms_avg"],
                        "planted_coef_max_err": r["planted_coef_max_err"]})
            for k in ("cpu_baseline", "cpu_baseline_port", "solve_ms_avg", "exchange_us", "fit_metrics"):
                if k in r:
                    out[k] = r[k]
        else:
            out.setdefault("also", {})[name] = r
    if args.workload in ("all", "gram_family"):
        fam = run_gram_family(rank, world)
        if args.workload == "gram_family":
            best = max(fam["shapes"].values(), key=lambda v: v["gram_frac"])
            out.update({"metric": "regression rows/s (multipleRegression family F = 2p, P = p)",
                        "value": best["rows"] / best["gram_ms"] * 1e3 * world, "ms_per_step": best["gram_ms"],
                        "config": fam["config"], "roofline": {
                            "bound": "hbm", "achieved": best["gram_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": best["gram_frac"], "traffic": None, "kernel": best["gram_kernel"]},
                        "shapes": fam["shapes"]})
        else:
            out.setdefault("also", {})["gram_family"] = fam
    if args.workload in ("all", "fit_walk"):
        fw = run_fit_walk(rank, world)
        if args.workload == "fit_walk":
            v1 = fw["shapes"]["V1"]
            out.update({"metric": "regression rows/s (multipleRegression epoch over a resident volume, "
                                  "Gram + metrics passes reading the volume)",
                        "value": v1["rows_per_s_walk"], "ms_per_step": v1["epoch_ms_walk"],
                        "config": fw["config"], "roofline": {
                            "bound": "hbm", "achieved": v1["gram_walk_GBs"], "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": v1["gram_walk_frac"], "traffic": None,
                            "kernel": "gram_walk_kernel"},
                        "shapes": fw["shapes"]})
        else:
            out.setdefault("also", {})["fit_walk"] = fw
    if args.workload in ("all", "gram64"):
        # config 5: 1e10 rows over 8 GPUs = 1.25e9 rows per GPU; a pass takes ~0.25 s, so the default
        # run keeps the full per-GPU size but caps the step count
        rows64 = args.rows64
        k64 = args.steps if args.workload == "gram64" else min(args.steps, 6)
        w64 = args.warmup if args.workload == "gram64" else min(args.warmup, 1)
        g64 = run_gram64(k64, w64, rank, world, cpu_base, rows64)
        g64["steps"], g64["warmup"] = k64, w64
        if args.workload == "gram64":
            out.update({"metric": "regression rows/s (64-feature least squares, rows generated "
                                  "per chunk, Gram build + solve)",
                        "value": g64["value"], "ms_per_step": g64["ms_per_step"],
                        "roofline": g64["roofline"], "config": g64["config"],
                        "kernel_rows_per_s": g64["kernel_rows_per_s"],
                        "planted_coef_rel_err": g64["planted_coef_rel_err"],
                        "exchange_us": g64["exchange_us"]})
            for k in ("cpu_baseline", "cpu_baseline_port"):
                if k in g64:
                    out[k] = g64[k]
        else:
            out.setdefault("also", {})["gram64"] = g64
    if args.workload in ("all", "conv3x3"):
        c = run_conv(args.steps, args.warmup, rank, world, cpu_base)
        if args.workload == "conv3x3":
            out.update({"metric": "grid-pts/s conv fwd+bwd", "unit": "grid-pts/s",
                        "value": c["value"], "ms_per_step": c["ms_per_step"],
                        "roofline": c["roofline"], "config": c["config"],
                        "exchange_us": c["exchange_us"]})
            for k in ("cpu_baseline", "cpu_baseline_port", "with_bwd_data", "deferred_finalize",
                      "fwd_ms", "bwd_wb_ms", "pair_ms"):
                if k in c:
                    out[k] = c[k]
        else:
            out.setdefault("also", {})["conv3x3"] = c
    if args.workload in ("all", "model32"):
        mres = run_model(args.steps, args.warmup, rank, world, cpu_base)
        if args.workload == "model32":
            out.update({"metric": "grid-pts/s model fwd+bwd", "unit": "grid-pts/s",
                        "value": mres["value"], "ms_per_step": mres["ms_per_step"],
                        "roofline": mres["roofline"], "config": mres["config"],
                        "exchange_us": mres["exchange_us"]})
            for k in ("cpu_baseline", "cpu_baseline_port"):
                if k in mres:
                    out[k] = mres[k]
        else:
            out.setdefault("also", {})["model32"] = mres
    if args.workload in ("all", "conv32"):
        cres = run_conv32(args.steps, args.warmup, rank, world, cpu_base)
        if args.workload == "conv32":
            out.update({"metric": "grid-pts/s conv32 fwd+bwd", "unit": "grid-pts/s",
                        "value": cres["value"], "ms_per_step": cres["ms_per_step"],
                        "roofline": cres["roofline"], "config": cres["config"],
                        "kernel_ms": cres["kernel_ms"], "exchange_us": cres["exchange_us"]})
            if cres.get("cpu_baseline"):
                out["cpu_baseline"] = cres["cpu_baseline"]
        else:
            out.setdefault("also", {})["conv32"] = cres
    if args.workload in ("all", "conv32x3"):
        k3 = args.steps if args.workload == "conv32x3" else min(args.steps, 8)
        w3 = args.warmup if args.workload == "conv32x3" else min(args.warmup, 2)
        sres = run_conv32x3(k3, w3, rank, world, cpu_base)
        sres["steps"], sres["warmup"] = k3, w3
        if args.workload == "conv32x3":
            out.update({"metric": "grid-pts/s 3-layer conv32 model fwd+bwd", "unit": "grid-pts/s",
                        "value": sres["value"], "ms_per_step": sres["ms_per_step"],
                        "roofline": sres["roofline"], "config": sres["config"]})
            if sres.get("cpu_baseline"):
                out["cpu_baseline"] = sres["cpu_baseline"]
        else:
            out.setdefault("also", {})["conv32x3"] = sres
    if args.workload in ("all", "fields"):
        fres = run_fields(args.steps, args.warmup, rank, world, cpu_base)
        if args.workload == "fields":
            out.update({"metric": "elements/s field statistics", "unit": "elements/s",
                        "value": fres["value"], "ms_per_step": fres["ms_per_step"],
                        "roofline": fres["stats"]["roofline"], "config": fres["config"],
                        "normalise": fres["normalise"], "gather": fres["gather"],
                        "cubes": fres["cubes"]})
            if "cpu_baseline" in fres:
                out["cpu_baseline"] = fres["cpu_baseline"]
        else:
            out.setdefault("also", {})["fields"] = fres
    if args.workload in ("all", "conv3d_model"):
        c3 = run_conv3d_model(args.steps, args.warmup, world=world)
        if args.workload == "conv3d_model":
            out.update({"metric": "samples/s Conv3D model fit step", "unit": "samples/s",
                        "value": c3["value"], "ms_per_step": c3["ms_per_step"],
                        "roofline": c3["roofline"], "config": c3["config"]})
        else:
            out.setdefault("also", {})["conv3d_model"] = c3
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
