"""The upsampling convolutions (nearest 2x folded into the gather) per tile: the general tiles that took them until round 5 against the ping-pong tiles (gemm_pp.hpp CONV == 2).
usage: python3 tools/conv_upsample_bench.py [reps]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels  # noqa: E402

L = _lib.lib()
vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CASES = [("UNet SDXL b4 32->64, 1280 ch", 8, 32, 32, 1280, 1280), ("UNet SDXL b4 64->128, 640 ch", 8, 64, 64, 640, 640), ("VAE b4 128->256, 512 ch", 4, 128, 128, 512, 512),
         ("VAE b4 256->512, 512 ch", 4, 256, 256, 512, 512), ("VAE b4 512->1024, 256 ch", 4, 512, 512, 256, 256), ("UNet SD1.5 b1 32->64, 640 ch", 2, 32, 32, 640, 640)]
VARIANTS = [9, 16, 17, 21, 18, 20]
rng = np.random.default_rng(0)
for name, n, h, w, cin, cout in CASES:
    A = rng.standard_normal((n, h, w, cin)).astype(np.float16)
    W = (rng.standard_normal((cout, 9 * cin)) / np.sqrt(9 * cin)).astype(np.float16)
    oh, ow = 2 * h, 2 * w
    M, N, K = n * oh * ow, cout, 9 * cin
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 4)
    line, first = [], None
    for v in VARIANTS:
        a = kernels.GemmArgs(A=dA.ptr, lda=cin, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=dC.ptr, ldc32=N, conv=1, n_img=n, H=h, W=w, Cin=cin, OH=oh, OW=ow, KH=3, KW=3, stride=1, pad=1,
                             upsample=1, tile_variant=v + 1)
        lab = kernels.gemm_variant(a)
        kernels.gemm(a)
        got = dC.download((M, N), np.float32)
        if first is None:
            first = got
        err = float(np.abs(got - first).max() / np.abs(first).max())
        ev = [vp(), vp()]
        for e in ev:
            L.mlsd_event_create(ctypes.byref(e))
        best = 1e9
        for _ in range(3):
            L.mlsd_event_record(ev[0], None)
            for _ in range(reps):
                kernels.gemm(a)
            L.mlsd_event_record(ev[1], None)
            L.mlsd_event_sync(ev[1])
            ms = ctypes.c_float()
            L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
            best = min(best, ms.value / reps)
        line.append(f"{lab.replace('gemm<', '').replace('>', ''):>22s} {best * 1e3:8.1f} us {2.0 * M * N * K / best / 1e9:7.1f} TF/s (diff {err:.1e})")
    print(f"{name:32s} {M}x{N}x{K}: " + " | ".join(line), flush=True)
