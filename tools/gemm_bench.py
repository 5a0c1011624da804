"""GEMM/conv micro-benchmark: every tile variant on the shapes that dominate the SDXL UNet (diagnostics)."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels  # noqa: E402

L = _lib.lib()
vp = _lib.vp


def time_gemm(M, N, K, variant, mode=0, reps=20, check=False, conv=None):
    rng = np.random.default_rng(0)
    L.mlsd_gemm_force_variant(variant)
    L.mlsd_gemm_set_panel(mode)
    if conv:
        n, h, w, cin, cout, k = conv
        A = rng.standard_normal((n, h, w, cin)).astype(np.float16)
        W = (rng.standard_normal((cout, k * k * cin)) / np.sqrt(k * k * cin)).astype(np.float16)
        M, N, K = n * h * w, cout, k * k * cin
    else:
        A = rng.standard_normal((M, K)).astype(np.float16)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 4)
    a = kernels.GemmArgs(A=dA.ptr, lda=K if not conv else conv[3], W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=dC.ptr, ldc32=N)
    if conv:
        n, h, w, cin, cout, k = conv
        a.conv, a.n_img, a.H, a.W, a.Cin, a.OH, a.OW, a.KH, a.KW, a.stride, a.pad = 1, n, h, w, cin, h, w, k, k, 1, k // 2
    kernels.gemm(a)
    if check:
        got = dC.download((M, N), np.float32)
        ref = A.reshape(M, -1).astype(np.float32) @ W.astype(np.float32).T
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 1e-4, (variant, err)
    ev = [vp(), vp()]
    for e in ev:
        L.mlsd_event_create(ctypes.byref(e))
    for _ in range(3):
        kernels.gemm(a)
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps):
        kernels.gemm(a)
    L.mlsd_event_record(ev[1], None)
    L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float()
    L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    t = ms.value / reps
    return t, 2.0 * M * N * K / t / 1e9


shapes = [(8192, 1280, 1280), (8192, 3840, 1280), (8192, 10240, 1280), (8192, 1280, 5120), (32768, 640, 640), (32768, 1920, 640),
          (32768, 5120, 640), (32768, 640, 2560), (616, 2560, 2048), (6528, 1280, 1280), (16384, 4096, 4096)]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2, 3, 4, 5]
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for v in variants:
    time_gemm(300, 200, 136, v, mode=mode, check=True)
print("shape".ljust(24) + "".join(f"v{v:<16d}" for v in variants))
for (M, N, K) in shapes:
    row = f"{M}x{N}x{K}".ljust(24)
    for v in variants:
        t, tf = time_gemm(M, N, K, v, mode=mode)
        row += f"{t * 1e3:6.0f}us {tf:5.0f}TF    "
    print(row)
print("conv 3x3 (n,h,w,cin,cout)")
for cv in [(8, 128, 128, 320, 320, 3), (8, 64, 64, 640, 640, 3), (8, 32, 32, 1280, 1280, 3), (8, 32, 32, 2560, 1280, 3)]:
    row = str(cv).ljust(24)
    for v in variants:
        t, tf = time_gemm(0, 0, 0, v, mode=mode, conv=cv)
        row += f"{t * 1e3:6.0f}us {tf:5.0f}TF    "
    print(row)
