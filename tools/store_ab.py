"""A/B of the widened fp16 epilogue stores of the ping-pong GEMMs (debug bit 32 = 8-byte pieces), interleaved repetitions."""
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
for kind, M, N, K, v in [("geglu", 8192, 10240, 1280, 18), ("f16", 8192, 3840, 1280, 18), ("f16", 8192, 1280, 1280, 19), ("geglu", 32768, 5120, 640, 18), ("f16", 32768, 1920, 640, 18), ("f16", 32768, 640, 640, 19)]:
    A = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16)); W = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    b = _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    nout = N // 2 if kind == "geglu" else N
    y = _lib.DeviceBuffer(M * nout * 2)
    a = kernels.GemmArgs(A=A.ptr, lda=K, W_=W.ptr, ldb=K, M=M, N=N, K=K, bias=b.ptr, C16=y.ptr, ldc16=nout, tile_variant=v, act=kernels.ACT_GEGLU if kind == "geglu" else 0)
    for _ in range(30): kernels.gemm(a)          # warm state
    t = {0: [], 32: []}; outs = {}
    for rep in range(4):
        for dbg in (32, 0):
            L.mlsd_gemm_set_debug(dbg); t[dbg].append(run(a))
            if rep == 0: outs[dbg] = y.download((M, nout), np.float16)
    L.mlsd_gemm_set_debug(0)
    print(f"{kind:6s} {M}x{N}x{K} {kernels.gemm_variant(a)}: 8-byte pieces {min(t[32]):7.1f} us (runs {' '.join(f'{x:.1f}' for x in t[32])}) | 16-byte {min(t[0]):7.1f} us (runs {' '.join(f'{x:.1f}' for x in t[0])}) | identical output: {np.array_equal(outs[0], outs[32])}")
