"""scda_amd -- MI355X (gfx950) implementation of the SCDA Faster-R-CNN hot path.

Layout
  csrc/         hand-written HIP kernels + the C ABI (include/scda_ops.h) -> libscda_ops.so
  native.py     ctypes binding of that C ABI (the only way Python reaches the kernels)
  dropin/       host-side mirror of the reference's operator/module interface
                (top-level packages `extensions`, `models`, `functions`, `utils`)

The product path never falls back to a CPU implementation: if libscda_ops.so is
missing or no HIP device is present, calling an operator raises.
"""
__version__ = "0.1.0"
