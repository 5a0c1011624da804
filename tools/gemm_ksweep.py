"""Fixed cost vs per-K-tile cost of the single-round 128x320 ping-pong GEMM (M x 1280 outputs: one tile per CU).
usage: python3 tools/gemm_ksweep.py [M] [N]      (diagnostics; prints a line fit t = t0 + nkt * dt per epilogue flavour)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib, kernels

L = _lib.lib(); vp = _lib.vp
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
VAR = int(sys.argv[3]) if len(sys.argv) > 3 else 18
rng = np.random.default_rng(0)
KS = [192, 320, 640, 1280, 2560, 5120]
Kmax = max(KS)
A = rng.standard_normal((M, Kmax)).astype(np.float16)
W = (rng.standard_normal((N, Kmax)) / np.sqrt(Kmax)).astype(np.float16)
R = rng.standard_normal((M, N)).astype(np.float32)
dA, dW, dR = _lib.from_numpy(A), _lib.from_numpy(W), _lib.from_numpy(R)
dC32, dC16 = _lib.DeviceBuffer(M * N * 4), _lib.DeviceBuffer(M * N * 2)
ev = [vp(), vp()]
for e in ev:
    L.mlsd_event_create(ctypes.byref(e))


def run(K, flavour, dbg, reps=30):
    a = kernels.GemmArgs(A=dA.ptr, lda=Kmax, W_=dW.ptr, ldb=Kmax, M=M, N=N, K=K)
    if flavour in ("f32", "f32+res"):
        a.C32, a.ldc32 = dC32.ptr, N
    if flavour in ("f16",):
        a.C16, a.ldc16 = dC16.ptr, N
    if flavour == "f32+res":
        a.resid, a.ldr = dR.ptr, N
    L.mlsd_gemm_force_variant(VAR)
    L.mlsd_gemm_set_debug(dbg)
    for _ in range(3):
        kernels.gemm(a)
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps):
        kernels.gemm(a)
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps * 1e3


print(f"# M={M} N={N} variant {VAR}: us per launch (back-to-back launches of the same GEMM)")
for flavour in ("f16", "f32", "f32+res"):
    for dbg in (0, 2, 1):
        ts = [run(K, flavour, dbg) for K in KS]
        nkt = np.array(KS) / 64
        dt, t0 = np.polyfit(nkt[2:], np.array(ts)[2:], 1)
        print(f"{flavour:8s} { {0: 'full       ', 1: 'no-epilogue', 2: 'generic-epi'}[dbg]} " + " ".join(f"K={K}:{t:6.1f}" for K, t in zip(KS, ts)) +
              f"   fit: t0 = {t0:5.1f} us, {dt:5.3f} us/K-tile ({2.0 * M * N * 64 / dt / 1e6:6.0f} TFLOP/s in the loop)")
L.mlsd_gemm_set_debug(0); L.mlsd_gemm_force_variant(0)
