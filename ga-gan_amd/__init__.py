"""MI355X-native hot path of StyleGAN2 few-shot domain adaptation.

Layout:
  csrc/                 HIP kernels for gfx950 + the C ABI (include/sg_ops.h)
  _lib.py               ctypes binding of the C ABI (fails loudly if the library is missing)
  upfirdn2d_plugin.py   the two native entry points with the reference plugins' signatures
  bias_act_plugin.py
  torch_utils/ops/      op API used by the networks (same names and signatures as the reference)
  training/             networks, loss and training step on top of the op API
"""
__version__ = '0.1.0'
