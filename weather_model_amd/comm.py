"""The library's own exchange step (include/wm.h: wm_comm_*, wm_allreduce_sum): RCCL through
libwm.so for hosts that do not use torch.distributed.  `weather_model_amd/dp.py` (torch.distributed,
backend nccl = RCCL) stays the default; this is the C-ABI route of SURVEY.md 8(b).

    id = comm.unique_id()            # rank 0; ship the 128 bytes to every rank
    comm.init(rank, nranks, id)      # collective, with this rank's GPU current
    comm.allreduce_sum_(acc)         # in place, on the current stream
    comm.destroy()
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

ID_BYTES = 128
_DTYPES = {torch.float32: 0, torch.float64: 1}


def unique_id() -> bytes:
    buf = ctypes.create_string_buffer(ID_BYTES)
    _lib.check(_lib.lib().wm_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
    return buf.raw


def init(rank: int, nranks: int, uid: bytes) -> None:
    if len(uid) != ID_BYTES:
        raise ValueError(f"the unique id is {ID_BYTES} bytes, got {len(uid)}")
    buf = ctypes.create_string_buffer(uid, ID_BYTES)
    _lib.check(_lib.lib().wm_comm_init(int(rank), int(nranks), ctypes.cast(buf, ctypes.c_void_p)))


def size() -> int:
    return int(_lib.lib().wm_comm_size())


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """Sum over ranks of one contiguous CUDA float32 / float64 buffer, in place, asynchronous on the
    current stream."""
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("all-reduce needs one contiguous CUDA buffer (do not fragment it)")
    if t.dtype not in _DTYPES:
        raise TypeError(f"float32 or float64, got {t.dtype}")
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().wm_allreduce_sum(t.data_ptr(), t.numel(), _DTYPES[t.dtype],
                                               torch.cuda.current_stream().cuda_stream))
    return t


def destroy() -> None:
    _lib.check(_lib.lib().wm_comm_destroy())
