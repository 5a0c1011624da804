"""fp32 output of a small-M linear launch against float64 numpy for K split 1 / 5 / 10 / 20 (same fp16 operands): is a deep split numerically different?"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
rng = np.random.default_rng(0)
for (M, N, Kd) in [(256, 1280, 1280), (64, 1280, 5120), (16, 1280, 1280), (256, 1280, 11520)]:
    A = rng.standard_normal((M, Kd)).astype(np.float16); W = (rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W); dC = _lib.DeviceBuffer(M * N * 4)
    ws = _lib.DeviceBuffer(128 << 20); fl = _lib.from_numpy(np.zeros(4096, np.uint32))
    out = []
    for variant, ks in [(1, 1), (1, 5), (1, 10), (1, 20), (0, 1), (0, 20), (29, 10)]:
        a = kernels.GemmArgs(A=dA.ptr, lda=Kd, W_=dW.ptr, ldb=Kd, M=M, N=N, K=Kd, C32=dC.ptr, ldc32=N, tile_variant=variant + 1, ksplit=ks, ws=ws.ptr, ws_bytes=128 << 20, sk_flags=fl.ptr)
        try:
            kernels.gemm(a); kernels.sync()
            c = dC.download((M, N), np.float32).astype(np.float64)
            out.append(f"{kernels.gemm_variant(a)}: {np.linalg.norm(c - ref) / np.linalg.norm(ref):.2e}")
        except Exception as e:
            out.append(f"v{variant} k/{ks}: {e}")
    print(f"{M}x{N}x{Kd}: " + " | ".join(out), flush=True)
