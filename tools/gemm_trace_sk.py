"""Phase stamps of a stream-K launch on the 128x320 (variant 28) or 256x256 (19) tile: prologue, K loop of the block's units, hand-off (publish or gather + epilogue), exit.
usage: python3 tools/gemm_trace_sk.py M N K [variant]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
L.mlsd_gemm_streamk_ws_bytes.restype = ctypes.c_size_t
M, N, K = [int(x) for x in sys.argv[1:4]]
VAR = int(sys.argv[4]) if len(sys.argv) > 4 else 28
DBG = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.default_rng(0)
dA = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
dC = _lib.DeviceBuffer(M * N * 4)
ws = _lib.DeviceBuffer(L.mlsd_gemm_streamk_ws_bytes()); fl = _lib.from_numpy(np.zeros(4096, np.uint32))
tb = _lib.DeviceBuffer(256 * 8 * 8)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
L.mlsd_gemm_set_debug(DBG)
for v in (VAR, 18 if VAR == 28 else 17):
    _lib.check(L.mlsd_memset(vp(tb.ptr), 0, ctypes.c_size_t(256 * 8 * 8), None)); L.mlsd_device_sync()
    a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=dC.ptr, ldc32=N, tile_variant=v + 1, ws=ws.ptr, ws_bytes=ws.nbytes, sk_flags=fl.ptr)
    for _ in range(3): kernels.gemm(a)
    L.mlsd_gemm_set_trace(vp(tb.ptr))
    L.mlsd_event_record(ev[0], None); kernels.gemm(a); L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    L.mlsd_gemm_set_trace(None)
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    t = tb.download((256, 8), np.uint64).astype(np.int64)
    t = t[t[:, 6] > t[:, 0]]
    med = lambda x: float(np.median(x)) / 100.0      # s_memtime ticks of 10 ns -> us
    print(f"{kernels.gemm_variant(a)} {M}x{N}x{K}: event {ms.value * 1e3:.1f} us; {len(t)} blocks stamped; median block life {med(t[:, 6] - t[:, 0]):.1f} us (max {np.max(t[:, 6] - t[:, 0]) / 100:.1f})")
    print(f"   prologue {med(t[:, 1] - t[:, 0]):.2f} | K loop {med(t[:, 2] - t[:, 1]):.2f} | publish + drain {med(t[:, 4] - t[:, 2]):.2f} (max {np.max(t[:, 4] - t[:, 2]) / 100:.2f}) | "
              f"arrivals {med(t[:, 5] - t[:, 4]):.2f} (max {np.max(t[:, 5] - t[:, 4]) / 100:.2f}) | combine + stores {med(t[:, 3] - t[:, 5]):.2f} (max {np.max(t[:, 3] - t[:, 5]) / 100:.2f}) | -> exit {med(t[:, 6] - t[:, 3]):.2f}   [x100 shader clocks]")
        continue
    print(f"   prologue {med(t[:, 1] - t[:, 0]):.2f} us | K loop until the first hand-off / epilogue {med(t[:, 2] - t[:, 1]):.2f} us | that hand-off / epilogue {med(t[:, 3] - t[:, 2]):.2f} us (max {np.max(t[:, 3] - t[:, 2]) / 100:.2f})"
          f" | last epilogue issue {med(t[:, 5] - t[:, 4]):.2f} | -> exit {med(t[:, 6] - t[:, 5]):.2f} | tiles per block {t[:, 7].min()}..{t[:, 7].max()}")
