"""libra_amd — MI355X (gfx950) native implementation of Libra's vision-to-LLM hot path.

Layout (only what the path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/libra_hip.h) -> lib/liblibra_hip.so
  _lib.py    ctypes loader (fails loudly, no fallback);  kernels.py  typed wrappers
  vit_engine.py   forward/backward kernel schedule of the CLIP ViT
  clip/, libra/   host-side mirrors of the reference's libra/models/{clip,libra} module surface
"""
__version__ = "0.1.0"
