"""CPU, world size 2, gloo: the N>1 exchange path (row shards + one all-reduce of the partial Gram
/ of the flat gradient).  The per-rank arithmetic here is done by the oracle (no GPU in this
container); the sharding and the collective are the product's (weather_model_amd/dp.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from weather_model_amd import dp, synth
    from oracle import cpu as ocpu, model_ref
    # --- regression: row shards by global index, partial Gram, one all-reduce, solve everywhere
    n, F = 10007, 16
    start, cnt = dp.shard(n)
    X, Y, _ = synth.regression_problem(cnt, F, 1, seed=1, row_offset=start)
    acc = torch.from_numpy(ocpu.gram(X, Y))
    dp.reduce_gram_(acc)
    beta = ocpu.normal_solve(acc.numpy(), F, 1)
    # --- model: sample shards, gradient of the global mean loss
    B = 37
    s0, c0 = dp.shard(B)
    params = {k: synth.uniform(600 + i, shp, -0.3, 0.3)
              for i, (k, shp) in enumerate(model_ref.PARAM_SHAPES.items())}
    x = synth.uniform(7, (B, 2, 6))
    y = synth.uniform(8, (B, 6))
    _, _, g = model_ref.loss_and_grads(params, x[s0:s0 + c0], y[s0:s0 + c0])
    flat = np.concatenate([g[k].reshape(-1) for k in model_ref.PARAM_NAMES]) * (c0 / B)
    flat_t = torch.from_numpy(flat)
    dp.reduce_grads_(flat_t)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), acc=acc.numpy(), beta=beta, grads=flat_t.numpy(),
             shard=np.array([start, cnt]))
    dist.destroy_process_group()


def test_two_rank_gram_and_gradient_exchange(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from weather_model_amd import synth
    from oracle import cpu as ocpu, model_ref
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    # shards tile the rows
    assert r0["shard"][0] == 0 and r0["shard"][0] + r0["shard"][1] == r1["shard"][0]
    assert r1["shard"][0] + r1["shard"][1] == 10007
    # both ranks hold the same, full-data result
    assert np.array_equal(r0["acc"], r1["acc"]) and np.array_equal(r0["beta"], r1["beta"])
    X, Y, _ = synth.regression_problem(10007, 16, 1, seed=1)
    full = ocpu.gram(X, Y)
    assert np.abs(r0["acc"] - full).max() <= 1e-9 * np.abs(full).max()
    assert np.abs(r0["beta"] - ocpu.lstsq(X, Y)).max() < 1e-7
    params = {k: synth.uniform(600 + i, shp, -0.3, 0.3)
              for i, (k, shp) in enumerate(model_ref.PARAM_SHAPES.items())}
    _, _, g = model_ref.loss_and_grads(params, synth.uniform(7, (37, 2, 6)), synth.uniform(8, (37, 6)))
    ref = np.concatenate([g[k].reshape(-1) for k in model_ref.PARAM_NAMES])
    assert np.abs(r0["grads"] - ref).max() <= 1e-12 + 1e-9 * np.abs(ref).max()
    assert np.array_equal(r0["grads"], r1["grads"])


def test_shard_properties():
    from weather_model_amd import dp
    for n in (0, 1, 7, 100, 10**10):
        for w in (1, 2, 3, 8):
            parts = [dp.shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        dp.shard(10, 2, 2)
