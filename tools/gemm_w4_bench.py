"""One wave per SIMD (variant 26, gemm_w4.hip) against the ping-pong tiles (17: 256x256, 18: 128x320) on the UNet's linear shapes, interleaved rounds
in one process; outputs compared.  usage: python3 tools/gemm_w4_bench.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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


# (label, M, N, K, residual, output: 0 f32 1 f16 2 geglu-f16)
CASES = [("GEGLU 8192x10240x1280", 8192, 10240, 1280, 0, 2), ("QKV 8192x3840x1280 f16", 8192, 3840, 1280, 0, 1), ("GEGLU 32768x5120x640", 32768, 5120, 640, 0, 2),
         ("QKV 32768x1920x640 f16", 32768, 1920, 640, 0, 1), ("FF-out 8192x1280x5120 f32+res", 8192, 1280, 5120, 1, 0), ("FF-out 32768x640x2560 f32+res", 32768, 640, 2560, 1, 0),
         ("q-proj 8192x1280x1280 f16", 8192, 1280, 1280, 0, 1), ("square 8192^3 f16", 8192, 8192, 8192, 0, 1), ("4096x4096x16384 f32", 4096, 4096, 16384, 0, 0), ("square 8192^3 f32", 8192, 8192, 8192, 0, 0),
         ("8192x10240x1280 f32", 8192, 10240, 1280, 0, 0)]
if len(sys.argv) > 2: CASES = [c for c in CASES if sys.argv[2] in c[0]]
for label, M, N, Kd, res, out in CASES:
    A = rng.standard_normal((M, Kd)).astype(np.float16)
    W = (rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16)
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 4)
    dR = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)) if res else None
    dB = _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    def mk(v):
        a = kernels.GemmArgs(A=dA.ptr, lda=Kd, W_=dW.ptr, ldb=Kd, M=M, N=N, K=Kd, tile_variant=v + 1, bias=dB.ptr)
        if out == 1: a.C16, a.ldc16 = dC.ptr, N
        elif out == 2: a.C16, a.ldc16, a.act = dC.ptr, N // 2, kernels.ACT_GEGLU
        else: a.C32, a.ldc32 = dC.ptr, N
        if res: a.resid, a.ldr = dR.ptr, N
        return a
    outs, line = {}, []
    No = N // 2 if out == 2 else N
    for v in (17, 26, 18, 27):
        if v in (18, 27) and (N % 160 or M % 64 or out == 2): continue
        a = mk(v)
        ts = sorted(timeit(a) for _ in range(3))
        L.mlsd_memset(vp(dC.ptr), 0xff, ctypes.c_size_t(M * No * (2 if out else 4)), None)
        kernels.gemm(a)
        outs[v] = dC.download((M, No), np.float16 if out else np.float32).astype(np.float32)
        line.append(f"{kernels.gemm_variant(a).split('<')[1].split(',')[0]:>13s} {ts[0]:8.1f} us {2.0 * M * N * Kd / ts[0] / 1e6:7.1f} TF/s")
    d = max(np.abs(outs[26] - outs[17]).max(), np.abs(outs[27] - outs[18]).max() if 27 in outs else 0.0)
    if 28 in outs: line.append(f"rel |w4m32 - pp| {np.linalg.norm(outs[28] - outs[17]) / np.linalg.norm(outs[17]):.1e}")
    print(f"{label:32s} | " + " | ".join(line) + f" | max |w4 - pp| {d:.1e} (|pp| max {np.abs(outs[17]).max():.1f})", flush=True)
