"""simpledet_amd -- MI355X (gfx950) backend for SimpleDet's second-stage detection ops.

Only what the hot path needs lives here:
  csrc/            hand-written HIP kernels + the C ABI (include/simpledet_ops.h)
  _lib.py          ctypes binding generated from the header (fails loudly if the .so is missing)
  ops.py           torch-tensor harness (device pointers + current stream -> C ABI)
  contrib.py       the reference's operator names (mx.sym.contrib.* mirror, autograd-enabled)
  mxnet_plugin.py  MXNet CustomOp registration (imports mxnet lazily)
  dist.py          one-process-per-GPU data-parallel harness (RCCL all-reduce of gradients)
"""
__version__ = "0.1.0"
