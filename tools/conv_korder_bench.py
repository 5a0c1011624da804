"""Slab-ordered K for the implicit-GEMM convolutions ((cin / 64, kh, kw, 64) instead of (kh, kw, cin)): the 9 taps of a 64-channel slab are consecutive K tiles, so an
input pixel's slab is re-read within 9 K tiles instead of once per tap pass.  Timing (and, under rocprofv3 --pmc, FETCH_SIZE) of the SDXL conv shapes in both orders.
usage: python3 tools/conv_korder_bench.py [reps]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)
def timeit(fns):
    for f in fns: f()
    L.mlsd_event_record(ev[0], None)
    for i in range(reps): fns[i % len(fns)]()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps * 1e3
if not L.mlsd_has_experiments():
    sys.exit("needs a library built with `make EXPERIMENTS=1` (mlsd_gemm_set_korder is a no-op in the product build)")
for (n, h, w, cin, cout, variant) in [(8, 32, 32, 1280, 1280, 20), (8, 32, 32, 2560, 1280, 20), (8, 64, 64, 640, 640, 20), (8, 64, 64, 1920, 640, 20), (8, 128, 128, 320, 320, 18), (8, 128, 128, 640, 320, 20),
                                      (4, 256, 256, 512, 512, 20), (4, 512, 512, 256, 256, 20)]:
    M, N, K = n * h * w, cout, 9 * cin
    W = _lib.from_numpy((rng.standard_normal((cout, K)) / np.sqrt(K)).astype(np.float16))
    sets = []
    for s in range(3):
        sets.append((_lib.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float16)), _lib.DeviceBuffer(M * N * 4)))
    def args(s):
        a = kernels.GemmArgs(A=sets[s][0].ptr, lda=cin, W_=W.ptr, ldb=K, M=M, N=N, K=K, C32=sets[s][1].ptr, ldc32=N, tile_variant=variant + 1)
        a.conv, a.n_img, a.H, a.W, a.Cin, a.OH, a.OW, a.KH, a.KW, a.stride, a.pad = 1, n, h, w, cin, h, w, 3, 3, 1, 1
        return a
    out = []
    for ko in (0, 1, 0, 1):
        L.mlsd_gemm_set_korder(ko)
        t = timeit([(lambda a=args(s): kernels.gemm(a)) for s in range(3)])
        out.append(f"korder {ko}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s")
    L.mlsd_gemm_set_korder(0)
    print(f"{kernels.gemm_variant(args(0))} {M}x{N}x{K}: " + " | ".join(out), flush=True)
