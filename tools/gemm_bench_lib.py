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


