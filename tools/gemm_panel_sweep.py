"""Tile-order panel width (mlsd_gemm_set_panel: tile columns per panel, 0 = row-major) against launch time for the 256x256 ping-pong GEMMs of the SDXL plan.
usage: python3 tools/gemm_panel_sweep.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
for M, N, K, geglu in ((8192, 10240, 1280, 1), (8192, 3840, 1280, 0), (32768, 5120, 640, 1), (32768, 1920, 640, 0)):
    dA = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16)); dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    dC = _lib.DeviceBuffer(M * N * 2)
    a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C16=dC.ptr, ldc16=N // 2 if geglu else N, tile_variant=18)
    if geglu: a.act = kernels.ACT_GEGLU
    out = []
    for rnd in range(2):
        for gw in (1, 2, 4, 5, 8, 10, 16, 20, 0):
            L.mlsd_gemm_set_panel(gw)
            for _ in range(3): kernels.gemm(a)
            L.mlsd_event_record(ev[0], None)
            for _ in range(reps): kernels.gemm(a)
            L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
            ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
            if rnd: out.append(f"gw {gw}: {ms.value / reps * 1e3:.1f}")
    L.mlsd_gemm_set_panel(8)
    print(f"{kernels.gemm_variant(a)} {M}x{N}x{K}{' geglu' if geglu else ''} (us per launch): " + " | ".join(out), flush=True)
