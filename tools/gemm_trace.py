"""Where a ping-pong GEMM launch spends its time, from s_memtime stamps written by thread 0 of every block
(mlsd_gemm_set_trace): prologue, main loop, epilogue issue, store drain.
usage: python3 tools/gemm_trace.py M N K [variant] [flavour: f16|f32|f32+res|geglu]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib, kernels

L = _lib.lib(); vp = _lib.vp
M, N, K = [int(x) for x in sys.argv[1:4]]
VAR = int(sys.argv[4]) if len(sys.argv) > 4 else 18
flav = sys.argv[5] if len(sys.argv) > 5 else "f32+res"
rng = np.random.default_rng(0)
A = rng.standard_normal((M, K)).astype(np.float16)
W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
R = rng.standard_normal((M, N)).astype(np.float32)
dA, dW, dR = _lib.from_numpy(A), _lib.from_numpy(W), _lib.from_numpy(R)
dC32, dC16 = _lib.DeviceBuffer(M * N * 4), _lib.DeviceBuffer(M * N * 2)
a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K)
if flav.startswith("f32"):
    a.C32, a.ldc32 = dC32.ptr, N
elif flav == "geglu":
    dBias = _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    a.C16, a.ldc16, a.act, a.bias = dC16.ptr, N // 2, kernels.ACT_GEGLU, dBias.ptr
else:
    a.C16, a.ldc16 = dC16.ptr, N
if flav == "f32+res":
    a.resid, a.ldr = dR.ptr, N
L.mlsd_gemm_force_variant(VAR)
tb = _lib.DeviceBuffer(256 * 8 * 8)
ev = [vp(), vp()]
for e in ev:
    L.mlsd_event_create(ctypes.byref(e))
DBGS = [int(x) for x in os.environ.get('TRACE_DBG', '0,2,1').split(',')]
for dbg in DBGS:
    L.mlsd_gemm_set_debug(dbg)
    for _ in range(3):
        kernels.gemm(a)
    L.mlsd_gemm_set_trace(vp(tb.ptr))
    L.mlsd_event_record(ev[0], None)
    kernels.gemm(a)
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    L.mlsd_gemm_set_trace(None)
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    t = tb.download((256, 8), np.uint64).astype(np.int64)
    nb = min(256, (M + (255 if VAR == 17 else 127)) // (256 if VAR == 17 else 128) * ((N + (255 if VAR == 17 else 319)) // (256 if VAR == 17 else 320)))
    t = t[:nb]
    # the counters of the 8 XCDs are not aligned with each other: only differences inside one block mean anything
    life = np.median(t[:, 6] - t[:, 0])
    f = lambda x: f"{np.median(x):9.0f} ticks = {100.0 * np.median(x) / life:5.1f} % of a block's life (max {np.max(x):9.0f})"
    print(f"{flav} {M}x{N}x{K} variant {VAR} dbg {dbg}: event {ms.value * 1e3:.1f} us; median block life {life:.0f} ticks")
    print(f"   prologue                 : {f(t[:, 1] - t[:, 0])}")
    print(f"   main loop of first tile  : {f(t[:, 2] - t[:, 1])}")
    print(f"   first epilogue (issue)   : {f(t[:, 3] - t[:, 2])}")
    print(f"   last epilogue (issue)    : {f(t[:, 5] - t[:, 4])}")
    print(f"   last epilogue -> drained : {f(t[:, 6] - t[:, 5])}")
    print(f"   tiles/block {t[:, 7].min()}..{t[:, 7].max()}")
L.mlsd_gemm_set_debug(0); L.mlsd_gemm_force_variant(0)
