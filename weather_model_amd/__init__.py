"""weather_model_amd -- MI355X (gfx950) implementation of the akollgaa/weather-model hot path.

Only what the path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI declared in include/wm.h  (-> libwm.so)
  _lib.py    ctypes loader (fails loudly if libwm.so is missing; no CPU fallback)
  ops.py     operator layer over PyTorch-ROCm tensors (conv2d + autograd, Gram, solve)
  synth.py   host side of the counter-based synthetic data generator
  dropin/    same-named mirrors of the reference's Python modules for this path

Importing the package does not load libwm.so; the first op does.
"""
__version__ = "0.1.0"
