"""What the per-tile bias fetch costs the ping-pong GEMM epilogues: the same launches with and without a bias vector
(interleaved repetitions, HIP events)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)
def run(a, reps=20):
    for _ in range(3): kernels.gemm(a)
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): kernels.gemm(a)
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps * 1e3
for kind, M, N, K, v in [("geglu", 8192, 10240, 1280, 18), ("f16", 8192, 3840, 1280, 18), ("f32res", 8192, 1280, 1280, 19), ("f32res", 8192, 1280, 5120, 19)]:
    A = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16)); W = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    b = _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    nout = N // 2 if kind == "geglu" else N
    def mk(with_bias):
        a = kernels.GemmArgs(A=A.ptr, lda=K, W_=W.ptr, ldb=K, M=M, N=N, K=K, tile_variant=v)
        if with_bias: a.bias = b.ptr
        if kind == "f32res": a.C32, a.ldc32, a.resid, a.ldr = y.ptr, N, r.ptr, N
        else: a.C16, a.ldc16 = y.ptr, nout
        if kind == "geglu": a.act = kernels.ACT_GEGLU
        return a
    y = _lib.DeviceBuffer(M * nout * (4 if kind == "f32res" else 2)); r = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)) if kind == "f32res" else None
    a1, a0 = mk(True), mk(False)
    for _ in range(30): kernels.gemm(a1)
    t1, t0 = [], []
    for rep in range(4):
        t1.append(run(a1)); t0.append(run(a0))
    print(f"{kind:7s} {M}x{N}x{K} {kernels.gemm_variant(a1)}: with bias {min(t1):7.1f} us ({' '.join(f'{x:.1f}' for x in t1)}) | without {min(t0):7.1f} us ({' '.join(f'{x:.1f}' for x in t0)})")
