"""Two phases per K tile (variants 20 / 21) against four (18 / 17) on the UNet's GEMM / conv shapes, interleaved rounds in one process.
usage: python3 tools/gemm_phases.py [reps]"""
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


# (label, M, N, K, conv geometry or None, residual, output: 0 f32 1 f16 2 geglu-f16)
CASES = [("out-proj 8192x1280x1280 f32+res", 8192, 1280, 1280, None, 1, 0), ("q-proj 8192x1280x1280 f16", 8192, 1280, 1280, None, 0, 1),
         ("FF-out 8192x1280x5120 f32+res", 8192, 1280, 5120, None, 1, 0), ("QKV 8192x3840x1280 f16", 8192, 3840, 1280, None, 0, 1),
         ("GEGLU 8192x10240x1280", 8192, 10240, 1280, None, 0, 2), ("FF-out 32768x640x2560 f32+res", 32768, 640, 2560, None, 1, 0),
         ("out-proj 32768x640x640 f32+res", 32768, 640, 640, None, 1, 0), ("QKV 32768x1920x640 f16", 32768, 1920, 640, None, 0, 1),
         ("GEGLU 32768x5120x640", 32768, 5120, 640, None, 0, 2),
         ("conv 8192x1280x11520 f32+res", 0, 1280, 0, (8, 32, 32, 1280, 3), 1, 0), ("conv 32768x640x5760 f32+res", 0, 640, 0, (8, 64, 64, 640, 3), 1, 0),
         ("conv 131072x320x2880 f32+res", 0, 320, 0, (8, 128, 128, 320, 3), 1, 0), ("conv 131072x320x5760 f32", 0, 320, 0, (8, 128, 128, 640, 3), 0, 0),
         ("conv 32768x640x11520 f32", 0, 640, 0, (8, 64, 64, 1280, 3), 0, 0), ("conv 8192x1280x23040 f32", 0, 1280, 0, (8, 32, 32, 2560, 3), 0, 0)]
for label, M, N, Kd, cv, res, out in CASES:
    if cv:
        n, h, w, cin, k = cv
        M, Kd = n * h * w, k * k * cin
        A = rng.standard_normal((n, h, w, cin)).astype(np.float16)
    else:
        A = rng.standard_normal((M, Kd)).astype(np.float16)
    W = (rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16)
    dA, dW = _lib.from_numpy(A), _lib.from_numpy(W)
    dC = _lib.DeviceBuffer(M * N * 4)
    dR = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)) if res else None
    dB = _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    def mk(v):
        a = kernels.GemmArgs(A=dA.ptr, lda=cv[3] if cv else Kd, W_=dW.ptr, ldb=Kd, M=M, N=N, K=Kd, tile_variant=v + 1)
        if out == 1: a.C16, a.ldc16 = dC.ptr, N
        elif out == 2: a.C16, a.ldc16, a.act, a.bias = dC.ptr, N // 2, kernels.ACT_GEGLU, dB.ptr
        else: a.C32, a.ldc32 = dC.ptr, N
        if res: a.resid, a.ldr = dR.ptr, N
        if cv:
            a.conv, a.n_img, a.H, a.W, a.Cin, a.OH, a.OW, a.KH, a.KW, a.stride, a.pad = 1, n, h, w, cin, h, w, k, k, 1, k // 2
        return a
    outs, line = {}, []
    No = N // 2 if out == 2 else N
    for v in (17, 21, 18, 20, 22):
        if v in (18, 20, 22) and (N % 80 or M % 64 or out == 2): continue
        a = mk(v)
        ts = sorted(timeit(a) for _ in range(3))
        outs[v] = dC.download((M, No), np.float16 if out else np.float32).astype(np.float32)
        line.append(f"{kernels.gemm_variant(a).split('<')[1].split(',')[0]:>14s} {ts[0]:8.1f} us {2.0 * M * N * Kd / ts[0] / 1e6:7.1f} TF/s")
    d = max(np.abs(outs[21] - outs[17]).max(), max(np.abs(outs[v] - outs[18]).max() for v in (20, 22)) if 20 in outs else 0.0)
    print(f"{label:32s} | " + " | ".join(line) + f" | max diff of the schedules vs 17 / 18: {d:.1e}", flush=True)
