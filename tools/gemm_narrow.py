"""Narrow outputs (N = 128: the VAE decoder's full-resolution convolutions): the 256x128 ping-pong tile (variant 25) against the tiles
the table held for them (4: 256x128x32s3, 3: 256x128x64s2).  usage: python3 tools/gemm_narrow.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)


def timeit(a):
    for _ in range(2): kernels.gemm(a)
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): kernels.gemm(a)
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps * 1e3


# (label, (n, h, w, cin, k), N, residual)
CASES = [("conv 1048576x128x1152 f32+res (sdxl b4 /4 rows)", (1, 1024, 1024, 128, 3), 128, 1), ("conv 1048576x128x1152 f32", (1, 1024, 1024, 128, 3), 128, 0),
         ("conv 1048576x128x2304 f32", (1, 1024, 1024, 256, 3), 128, 0), ("conv 262144x128x1152 f32+res (sd15 b1)", (1, 512, 512, 128, 3), 128, 1),
         ("conv 262144x128x2304 f32", (1, 512, 512, 256, 3), 128, 0), ("conv 1048576x128x256 f32 (1x1)", (1, 1024, 1024, 256, 1), 128, 0)]
for label, (n, h, w, cin, k), N, res in CASES:
    M, Kd = n * h * w, k * k * cin
    A = rng.standard_normal((n, h, w, cin)).astype(np.float16)
    W = (rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16)
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 4)
    dR = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)) if res else None
    outs, line = {}, []
    for v in (4, 3, 25):
        a = kernels.GemmArgs(A=dA.ptr, lda=cin, W_=dW.ptr, ldb=Kd, M=M, N=N, K=Kd, tile_variant=v + 1, C32=dC.ptr, ldc32=N,
                             conv=1, n_img=n, H=h, W=w, Cin=cin, OH=h, OW=w, KH=k, KW=k, stride=1, pad=k // 2)
        if res: a.resid, a.ldr = dR.ptr, N
        ts = sorted(timeit(a) for _ in range(3))
        outs[v] = dC.download((M, N), np.float32)
        line.append(f"{kernels.gemm_variant(a).split('<')[1].split(',')[0]:>14s} {ts[0]:8.1f} us {2.0 * M * N * Kd / ts[0] / 1e6:7.1f} TF/s")
    d = np.abs(outs[25] - outs[4]).max()
    print(f"{label:50s} | " + " | ".join(line) + f" | max diff 25 vs 4: {d:.1e}", flush=True)
