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
