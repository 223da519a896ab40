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


def _fake_ops(monkey_ops):
    """CPU stand-ins for the two GPU launches DeviceTrainer makes, built on the oracle: this test is
    about the trainer's control flow under torch.distributed (collective pairing, batch counts,
    weight broadcast), which has no GPU in this container."""
    from oracle import model_ref

    def model_train(flat, x, y=None, batch_global=None, **kw):
        p, off = {}, 0
        for name in model_ref.PARAM_NAMES:
            shp = model_ref.PARAM_SHAPES[name]
            k = int(np.prod(shp))
            p[name] = flat[off:off + k].numpy().reshape(shp)
            off += k
        B = x.shape[0]
        loss, _, g = model_ref.loss_and_grads(p, x.numpy(), y.numpy())
        scale = B / float(batch_global or B)
        grads = np.concatenate([g[k].reshape(-1) for k in model_ref.PARAM_NAMES]) * scale
        return None, torch.tensor(loss * scale, dtype=torch.float32), \
            torch.from_numpy(grads.astype(np.float32))

    def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = v.sqrt() / (1 - b2 ** step) ** 0.5 + eps
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))

    monkey_ops.model_train = model_train
    monkey_ops.adam_step = adam_step


def _trainer_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from weather_model_amd import ops, synth
    from weather_model_amd.dropin import newConvolution as nc
    _fake_ops(ops)
    T, Z, Y, X = 9, 1, 3, 7                       # 3 * 1 * 3 * 7 = 63 samples
    data = {v: torch.from_numpy(synth.uniform(100 + k, (T, 1, Z, Y, X)))
            for k, v in enumerate(nc.data_vars)}
    shape = (T - 6, Z, Y, X)
    n = int(np.prod(shape))
    ds = nc.CustomDataset(data, n - 2, shape, 0)  # 61 samples: 61 % 4 == 1 -> a one-sample last batch
    torch.manual_seed(1234 + 17 * rank)           # every rank draws DIFFERENT initial weights
    model = nc.CustomModel((2, 6)).to("cpu")
    trainer = nc.DeviceTrainer(model)
    l0 = trainer.epoch(ds, batch_size=4, shuffle=True, seed=3)
    l1 = trainer.epoch(ds, batch_size=4, shuffle=False, max_batches=5)
    np.savez(os.path.join(out_dir, f"t{world}_{rank}.npz"), flat=trainer.flat.numpy(),
             losses=np.array([l0, l1]), step=np.array([trainer.step]),
             w0=model.conv1.weight.detach().numpy())
    if world > 1:
        dist.destroy_process_group()


def test_device_trainer_two_ranks_ragged_last_batch(tmp_path):
    """61 samples, batch 4, world 2: the last global batch has ONE sample, so rank 1's share of it is
    empty.  Both ranks must still meet in every collective, count the same batches and steps, start
    from rank 0's weights, and end on the weights a single process reaches on the same batches."""
    port = _free_port()
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "t2_0.npz")
    b = np.load(tmp_path / "t2_1.npz")
    assert int(a["step"][0]) == int(b["step"][0]) == 16 + 5      # ceil(61 / 4) batches, then 5
    assert np.array_equal(a["flat"], b["flat"])                  # replicas stay identical
    assert np.array_equal(a["w0"], b["w0"])                      # and are written back to the module
    assert np.allclose(a["losses"], b["losses"], rtol=0, atol=0)
    # a single process with rank 0's seed runs the same global batches
    _trainer_worker(0, 1, _free_port(), str(tmp_path))
    s = np.load(tmp_path / "t1_0.npz")
    assert int(s["step"][0]) == 21
    assert np.abs(a["flat"] - s["flat"]).max() <= 2e-6
    assert np.allclose(a["losses"], s["losses"], rtol=1e-5)
