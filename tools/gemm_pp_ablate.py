"""Ablation of the ping-pong tile's seam cost: dbg 0 = full, 1 = no epilogue, 2 = epilogue without global stores."""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
rng = np.random.default_rng(0)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
for (M, N, K) in [(8192, 10240, 1280), (8192, 10240, 5120), (8192, 10240, 320)]:
    A = rng.uniform(-1, 1, (M, K)).astype(np.float16); W = rng.uniform(-1, 1, (N, K)).astype(np.float16)
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 2)
    out = []
    for dbg in (0, 1, 2, 0):
        L.mlsd_gemm_set_debug(dbg)
        a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C16=dC.ptr, ldc16=N, tile_variant=18)
        for _ in range(3): kernels.gemm(a)
        L.mlsd_event_record(ev[0], None)
        for _ in range(10): kernels.gemm(a)
        L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
        ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
        out.append(f"dbg{dbg}: {ms.value/10*1e3:7.1f} us {2.0*M*N*K/(ms.value/10)/1e9:6.0f} TF")
    L.mlsd_gemm_set_debug(0)
    print(f"{M}x{N}x{K}".ljust(18), " | ".join(out))
