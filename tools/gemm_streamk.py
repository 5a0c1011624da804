"""Stream-K (tile variant 19) against the plain persistent tiles (17: 256x256, 18: 128x320) on the UNet's shapes whose 256x256 tiles do not
fill whole rounds of the 256 blocks; linear and conv, fp32 (+ residual) and fp16 outputs.  Interleaved rounds in one process.
usage: python3 tools/gemm_streamk.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L.mlsd_gemm_streamk_ws_bytes.restype = ctypes.c_size_t
ws = _lib.DeviceBuffer(L.mlsd_gemm_streamk_ws_bytes())
flags = _lib.DeviceBuffer(4096)
_lib.check(L.mlsd_memset(vp(flags.ptr), 0, ctypes.c_size_t(4096), None)); L.mlsd_device_sync()
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


# (label, M, N, K, conv geometry or None, residual, fp16 output)
CASES = [("FF-out 8192x1280x5120 f32+res", 8192, 1280, 5120, None, 1, 0), ("QKV 8192x3840x1280 f16", 8192, 3840, 1280, None, 0, 1),
         ("out-proj 8192x1280x1280 f32+res", 8192, 1280, 1280, None, 1, 0), ("QKV 32768x1920x640 f16", 32768, 1920, 640, None, 0, 1),
         ("conv 8192x1280x11520 f32+res", 0, 1280, 0, (8, 32, 32, 1280, 3), 1, 0), ("conv 8192x1280x23040 f32", 0, 1280, 0, (8, 32, 32, 2560, 3), 0, 0),
         ("conv 8192x1280x17280 f32", 0, 1280, 0, (8, 32, 32, 1920, 3), 0, 0), ("conv 32768x1280x11520 f32", 0, 1280, 0, (8, 64, 64, 1280, 3), 0, 0),
         ("conv 32768x640x5760 f32+res", 0, 640, 0, (8, 64, 64, 640, 3), 1, 0), ("conv 131072x640x5760 f32", 0, 640, 0, (8, 128, 128, 640, 3), 0, 0)]
for label, M, N, Kd, cv, res, f16 in CASES:
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
    def mk(v):
        a = kernels.GemmArgs(A=dA.ptr, lda=cv[3] if cv else Kd, W_=dW.ptr, ldb=Kd, M=M, N=N, K=Kd, tile_variant=v + 1,
                             ws=ws.ptr, ws_bytes=ws.nbytes, sk_flags=flags.ptr)
        if f16: a.C16, a.ldc16 = dC.ptr, N
        else: a.C32, a.ldc32 = dC.ptr, N
        if res: a.resid, a.ldr = dR.ptr, N
        if cv:
            a.conv, a.n_img, a.H, a.W, a.Cin, a.OH, a.OW, a.KH, a.KW, a.stride, a.pad = 1, n, h, w, cin, h, w, k, k, 1, k // 2
        return a
    outs, line = {}, []
    for v in (17, 18, 19):
        if v == 18 and (N % 80 or M % 64): continue
        a = mk(v)
        name = kernels.gemm_variant(a)
        ts = sorted(timeit(a) for _ in range(3))
        outs[v] = dC.download((M, N), np.float16 if f16 else np.float32).astype(np.float32)
        line.append(f"{name.split('<')[1].split(',')[0]:>15s} {ts[0]:8.1f} us {2.0 * M * N * Kd / ts[0] / 1e6:7.1f} TF/s")
    d = np.abs(outs[19] - outs[17]).max() / max(np.abs(outs[17]).max(), 1e-30)
    print(f"{label:34s} | " + " | ".join(line) + f" | max rel diff sk vs 17: {d:.1e}", flush=True)
