"""GPU: the C-ABI exchange step (wm_comm_*, wm_allreduce_sum over RCCL) at world size 1 -- all a
one-GPU lease can run (RCCL refuses two ranks on one device).  A one-rank sum all-reduce must leave
the buffer bit for bit, on the stream it was given, for both dtypes; the communicator life cycle
and the error states are exercised.  The N > 1 path is the same single ncclAllReduce call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_world_one_allreduce_and_life_cycle():
    from weather_model_amd import comm, ops, _lib
    comm.destroy()
    assert comm.size() == 0
    uid = comm.unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    comm.init(0, 1, uid)
    try:
        assert comm.size() == 1
        with pytest.raises(_lib.WmError, match="already holds"):
            comm.init(0, 1, uid)
        acc = ops.synth_uniform((66, 66), seed=3).double()          # the F = 64 Gram's shape
        ref = acc.clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            comm.allreduce_sum_(acc)
        side.synchronize()
        assert torch.equal(acc, ref)
        g = ops.synth_uniform((20566,), seed=4, lo=-1.0, hi=1.0)    # the model's flat gradient
        gref = g.clone()
        comm.allreduce_sum_(g)
        torch.cuda.synchronize()
        assert torch.equal(g, gref)
        with pytest.raises(TypeError):
            comm.allreduce_sum_(g.half())
        with pytest.raises(ValueError):
            comm.allreduce_sum_(acc.t())
        # a Gram pass, the exchange, the solve: the sequence of INTEGRATION.md on the C-ABI route
        from weather_model_amd import synth
        X, Y, _ = synth.regression_problem(30000, 16, 1, seed=6)
        a = ops.gram(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda())
        comm.allreduce_sum_(a)
        beta = ops.normal_solve(a, 16, 1).cpu().numpy()
        from oracle import cpu as ocpu
        assert np.abs(beta - ocpu.lstsq(X, Y)).max() / np.abs(beta).max() < 1e-4
    finally:
        comm.destroy()
    assert comm.size() == 0
