"""Skinny-M weight-streaming kernel (tile variant 29, gemm_skinny.hpp) against the previous choices (64x128 / 128x128 split-K tiles) on the M <= 128 shapes of the SD1.5
batch-1 plan: per-launch time over COLD weights (a ring of weight copies larger than L2 + MALL, so that every launch streams its weights from HBM as it does in the plan),
HIP events around 20 launches.  Reports the weight bytes per second of the launch (+ its splitk_reduce).
usage: python3 tools/gemm_skinny.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
NCOPY = 24
rng = np.random.default_rng(0)


def bench(M, N, Kd, conv, variant, ksplit, reps=NCOPY):
    cin = Kd // 9 if conv else Kd
    A = _lib.from_numpy((rng.standard_normal((M, cin if conv else Kd))).astype(np.float16))
    Wh = (rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16)
    Ws = [_lib.from_numpy(Wh) for _ in range(NCOPY)]             # 24 x 29.5 MB = 708 MB > MALL
    C = _lib.DeviceBuffer(M * N * 4)
    nws = kernels.gemm_splitk_ws_bytes(max(M, 128), N, max(ksplit, 8))
    ws = _lib.DeviceBuffer(nws)
    args = []
    for w in Ws:
        a = kernels.GemmArgs(A=A.ptr, lda=cin if conv else Kd, W_=w.ptr, ldb=Kd, M=M, N=N, K=Kd, C32=C.ptr, ldc32=N, tile_variant=variant + 1, ksplit=ksplit, ws=ws.ptr, ws_bytes=nws)
        if conv:
            hw = (2, 8, 8) if M == 128 else (1, M, 1)
            a.conv, a.n_img, a.H, a.W, a.Cin, a.OH, a.OW, a.KH, a.KW, a.stride, a.pad = 1, hw[0], hw[1], hw[2], cin, hw[1], hw[2], 3, 3, 1, 1
        args.append(a)
    for a in args[:3]: kernels.gemm(a)
    best = 1e9
    for _ in range(3):
        L.mlsd_event_record(ev[0], None)
        for a in args: kernels.gemm(a)
        L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
        ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
        best = min(best, ms.value / len(args))
    name = kernels.gemm_variant(args[0])
    return best * 1e3, name


print(f"# {'shape':28s} {'kernel':44s} {'us/launch':>9s} {'weight TB/s':>11s}")
for (M, N, Kd, conv) in [(128, 1280, 11520, 1), (128, 1280, 23040, 1), (128, 1280, 2560, 0), (128, 1280, 1280, 0), (128, 1280, 5120, 0), (128, 3840, 1280, 0), (2, 20160, 1280, 0), (64, 1280, 11520, 1)]:
    cands = [(1, 13), (1, 8), (0, 26)] + [(29, s) for s in (6, 9, 13, 20, 26, 40)]
    if os.environ.get("SKINNY_DBG"):
        L.mlsd_gemm_set_debug(int(os.environ["SKINNY_DBG"]))
        cands = [(29, s) for s in (6, 9, 13, 20, 26)]
        if M != 128: continue
    for v, ks in cands:
        if ks > Kd // 128: continue
        try:
            us, name = bench(M, N, Kd, conv, v, ks)
        except Exception as e:
            print(f"{M}x{N}x{Kd}{' conv' if conv else ''}: variant {v} k/{ks} failed: {e}"); continue
        print(f"{(str(M) + 'x' + str(N) + 'x' + str(Kd) + (' conv' if conv else '')):30s} {name:44s} {us:9.1f} {N * Kd * 2 / us / 1e6:11.2f}", flush=True)
