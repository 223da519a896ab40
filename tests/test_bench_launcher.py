"""CPU: `python bench.py --gpus N` with no WORLD_SIZE starts N ranks itself and relays exactly one
JSON line (the driver's command shape); a failing rank makes the launcher fail."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=240)


def test_self_launch_two_ranks_plumbing():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "plumbing"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_ranks"] == 2 and out["rccl_ranks"] == 2 and out["launcher"] == "self"
    assert out["backend"] == "gloo" and out["steps"] == 2 and out["warmup"] == 1
    assert out["planted_coef_max_err"] < 1e-3


def test_single_rank_plumbing_needs_no_launcher():
    r = _run(["--steps", "2", "--warmup", "1", "--workload", "plumbing"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_ranks"] == 1 and out["launcher"] == "single" and out["backend"] is None


def test_gpu_workload_without_gpu_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "gram16"])
    assert r.returncode != 0
    assert "GPU" in r.stderr


def test_launcher_counts_gpus_without_hip(tmp_path, monkeypatch):
    """The launcher parent must never touch the GPU: devices come from the *_VISIBLE_DEVICES
    variables or from the KFD topology files (GPU nodes have simd_count > 0), not from
    torch.cuda.device_count(), which initialises HIP when amdsmi is absent."""
    sys.path.insert(0, ROOT)
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    for k, simd in enumerate((0, 0, 1024, 1024, 1024)):          # two CPU nodes, three GPUs
        d = tmp_path / str(k)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\n")
    assert bench.visible_gpus(str(tmp_path)) == 3
    assert bench.visible_gpus(str(tmp_path / "missing")) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert bench.visible_gpus(str(tmp_path)) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus(str(tmp_path)) == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src[src.index("def launch_ranks"):src.index("def dist_setup")]
    assert "torch.cuda" not in launcher
