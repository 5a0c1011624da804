"""Square/long-K GEMM timing with fp16 output only (main-loop comparison of tile variants in ONE process).
usage: python3 tools/gemm_square.py 9,17 [rounds]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
variants = [int(v) for v in sys.argv[1].split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 10240, 1280), (8192, 1280, 5120), (8192, 3840, 1280), (4096, 4096, 16384)]
rng = np.random.default_rng(0)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
for (M, N, K) in shapes:
    A = rng.uniform(-1, 1, (M, K)).astype(np.float16); W = rng.uniform(-1, 1, (N, K)).astype(np.float16)
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 2)
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C16=dC.ptr, ldc16=N, tile_variant=v + 1)
            for _ in range(2): kernels.gemm(a)
            L.mlsd_event_record(ev[0], None)
            for _ in range(10): kernels.gemm(a)
            L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
            ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
            res[v].append(2.0 * M * N * K / (ms.value / 10) / 1e9)
    print(f"{M}x{N}x{K}".ljust(20) + "  ".join(f"v{v}: med {np.median(res[v]):6.0f} max {max(res[v]):6.0f} TF" for v in variants))
