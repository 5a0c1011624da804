"""mlimgsynth_amd: MI355X-native Stable Diffusion denoising hot path.

Drop-in for the ggml op layer behind the reference's mlblock builder API
(reference: src/mlblock.h, src/mlblock_nn.h, src/unet.c, src/vae.c, src/tae.c,
src/clip.c, src/sampling.c, src/solvers.c).  The product is the C-ABI library
``lib/libmlimgsynth_amd.so`` (headers in ``include/``); this package is the thin
Python (ctypes) mirror used by tests, bench.py and the multi-GPU launcher.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
