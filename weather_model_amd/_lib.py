"""ctypes binding of libwm.so (include/wm.h).  Loading fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwm.so")

c_int, c_i64, c_u64 = ctypes.c_int, ctypes.c_int64, ctypes.c_uint64
c_f32, c_f64, c_sz, vp = ctypes.c_float, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/wm.h one to one (tests check the two agree)
SIGNATURES = {
    "wm_version": (c_int, []),
    "wm_last_error": (ctypes.c_char_p, []),
    "wm_device_cus": (c_int, []),
    "wm_conv2d_fwd_f32": (c_int, [vp, vp, vp, vp] + [c_int] * 7 + [c_int, vp]),
    "wm_conv2d_bwd_data_f32": (c_int, [vp, vp, vp] + [c_int] * 7 + [vp]),
    "wm_conv2d_bwd_data_relu_f32": (c_int, [vp, vp, vp, vp] + [c_int] * 7 + [vp]),
    "wm_relu_mask_f32": (c_int, [vp, vp, c_i64, vp]),
    "wm_relu_mask_to_f32": (c_int, [vp, vp, vp, c_i64, vp]),
    "wm_conv3d_bwd_data_relu_f32": (c_int, [vp, vp, vp, vp] + [c_int] * 9 + [vp]),
    "wm_conv2d_bwd_wb_workspace_bytes": (c_sz, [c_int] * 7),
    "wm_conv2d_bwd_wb_f32": (c_int, [vp, vp, vp, vp, vp, c_sz] + [c_int] * 7 + [vp]),
    "wm_conv2d_bwd_wb_partials_f32": (c_int, [vp, vp, vp, c_sz] + [c_int] * 7 + [vp, vp]),
    "wm_partials_finish_f32": (c_int, [vp, c_int, c_int, c_int, c_int, vp, vp, vp]),
    "wm_adam_step_partials_f32": (c_int, [vp, vp, c_int, c_int, vp, vp, c_int, vp, c_int, c_f64,
                                          c_f64, c_f64, c_f64, c_f32, vp]),
    "wm_conv2d_bwd_wb_noalloc_max_bytes": (c_sz, []),
    "wm_conv2d_bwd_wb_noalloc_f32": (c_int, [vp, vp, vp, vp] + [c_int] * 7 + [vp]),
    "wm_conv3d_fwd_f32": (c_int, [vp, vp, vp, vp] + [c_int] * 9 + [c_int, vp]),
    "wm_conv3d_bwd_data_f32": (c_int, [vp, vp, vp] + [c_int] * 9 + [vp]),
    "wm_conv3d_bwd_wb_workspace_bytes": (c_sz, [c_int] * 9),
    "wm_conv3d_bwd_wb_clock_offset": (c_sz, [c_int] * 9),
    "wm_conv3d_bwd_wb_f32": (c_int, [vp, vp, vp, vp, vp, c_sz] + [c_int] * 9 + [vp]),
    "wm_maxpool3d_fwd_f32": (c_int, [vp, vp, vp] + [c_int] * 5 + [vp]),
    "wm_maxpool3d_bwd_f32": (c_int, [vp, vp, vp] + [c_int] * 5 + [vp]),
    "wm_maxpool3d_fwd_pair_f32": (c_int, [vp] * 6 + [c_int] * 5 + [c_i64, vp]),
    "wm_maxpool3d_bwd_pair_f32": (c_int, [vp] * 6 + [c_int] * 5 + [c_i64, vp]),
    "wm_copy_multi": (c_int, [vp, c_int, vp]),
    "wm_gram_workspace_bytes": (c_sz, [c_i64, c_int, c_int]),
    "wm_gram_f32": (c_int, [vp, vp, c_i64, c_int, c_int, vp, vp, c_sz, vp]),
    "wm_normal_solve_workspace_bytes": (c_sz, [c_int, c_int]),
    "wm_normal_solve_f64": (c_int, [vp, c_int, c_int, c_int, c_f64, vp, vp, vp, c_sz, vp]),
    "wm_linear_predict_f32": (c_int, [vp, vp, vp, c_i64, c_int, c_int, vp]),
    "wm_linear_metrics_workspace_bytes": (c_sz, []),
    "wm_linear_metrics_supported": (c_int, [c_int, c_int]),
    "wm_linear_metrics_f32": (c_int, [vp, vp, vp, c_i64, c_int, c_int, c_int, c_f32, vp, vp, c_sz, vp]),
    "wm_residual_sums_f32": (c_int, [vp, vp, c_i64, c_int, c_f32, vp, vp, c_sz, vp]),
    "wm_linear_metrics_wide_supported": (c_int, [c_int, c_int]),
    "wm_linear_metrics_wide_workspace_bytes": (c_sz, [c_int, c_int]),
    "wm_linear_metrics_wide_f32": (c_int, [vp, vp, vp, c_i64, c_int, c_int, c_int, c_f32, vp, vp, vp, c_sz, vp]),
    "wm_model_param_count": (c_int, []),
    "wm_model_fwd_f32": (c_int, [vp, vp, vp, c_i64, vp]),
    "wm_model_workspace_bytes": (c_sz, [c_i64]),
    "wm_model_train_f32": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_sz, c_i64, c_i64, vp]),
    "wm_model_train_gather_f32": (c_int, [vp, vp, vp, c_i64] + [c_int] * 11 + [vp, vp, vp, vp, c_sz,
                                          c_i64, c_i64, vp]),
    "wm_adam_step_f32": (c_int, [vp, vp, vp, vp, c_i64, c_int, c_f64, c_f64, c_f64, c_f64, c_f32, vp]),
    "wm_adam_multi_f32": (c_int, [vp, c_int, c_i64, vp, c_f64, c_f64, c_f64, c_f64, c_int, c_int, vp]),
    "wm_field_stats_workspace_bytes": (c_sz, [c_i64]),
    "wm_field_stats_f32": (c_int, [vp, c_i64, vp, c_int, vp, c_sz, vp]),
    "wm_affine_scale_f32": (c_int, [vp, vp, c_i64, c_f32, c_f32, vp]),
    "wm_gather_triplets_f32": (c_int, [vp, vp, c_i64, vp, vp, c_i64] + [c_int] * 12 + [vp]),
    "wm_gather_cubes_f32": (c_int, [vp] + [c_int] * 10 + [vp, c_i64, c_int, c_i64, c_i64, c_i64,
                                    c_i64, c_i64, c_int, c_int, c_int, vp, vp, c_i64, vp, vp]),
    "wm_gather_cubes_strided_f32": (c_int, [vp] + [c_int] * 5 + [vp] + [c_int] * 5 + [vp, c_i64, c_int,
                                            c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int,
                                            vp, vp, c_i64, vp, vp]),
    "wm_gram_walk_workspace_bytes": (c_sz, [c_int]),
    "wm_gram_walk_f32": (c_int, [vp] + [c_int] * 5 + [vp, c_int, c_int, c_i64, c_int, c_i64, c_i64, c_i64,
                                 c_i64, c_i64, c_int, c_int, c_int, vp, vp, c_sz, vp]),
    "wm_linear_metrics_walk_f32": (c_int, [vp] + [c_int] * 5 + [vp, c_int, c_int, c_i64, c_int, c_i64,
                                           c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, vp, c_int,
                                           c_f32, vp, vp, c_sz, vp]),
    "wm_dense_fwd_f32": (c_int, [vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp]),
    "wm_dense_bwd_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "wm_dense_bwd_f32": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_sz, c_int, c_int, c_int, vp]),
    "wm_mse_metrics_workspace_bytes": (c_sz, [c_i64]),
    "wm_mse_metrics_f32": (c_int, [vp, vp, c_int, c_i64, c_int, c_f32, vp, vp, vp, c_sz, vp]),
    "wm_comm_unique_id": (c_int, [vp]),
    "wm_comm_init": (c_int, [c_int, c_int, vp]),
    "wm_comm_size": (c_int, []),
    "wm_allreduce_sum": (c_int, [vp, c_sz, c_int, vp]),
    "wm_comm_destroy": (c_int, []),
    "wm_synth_uniform_f32": (c_int, [vp, c_i64, c_u64, c_u64, c_f32, c_f32, vp]),
    "wm_synth_regression_f32": (c_int, [vp, vp, c_i64, c_int, c_int, c_u64, c_u64, vp, c_f32,
                                        c_f32, c_int, vp]),
}

_lib = None


class WmError(RuntimeError):
    """A libwm.so entry point returned a negative WM_ERR_* code."""

    def __init__(self, code, msg):
        super().__init__(f"libwm error {code}: {msg}")
        self.code = code


def lib() -> ctypes.CDLL:
    """Load libwm.so (built in-tree by weather_model_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  weather_model_amd has no CPU fallback.")
        # torch first: libwm.so needs libamdhip64.so.7 and must bind to the HIP runtime PyTorch
        # ships and initialises -- a second runtime loaded ahead of torch sees no device.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)     # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise WmError(rc, lib().wm_last_error().decode("utf-8", "replace"))
