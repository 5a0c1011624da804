"""Small-M GEMM study (SD1.5 batch 1 shapes): tile variants x K slices, HIP events per launch (operands hot in L2 / MALL).
VARIANTS can name an experimental tile index as well (round 2: a 4-deep ring of the 64x128 tile was measured with it and dropped)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)
ws = _lib.DeviceBuffer(128 << 20)
flags = _lib.from_numpy(np.zeros(4096, np.uint32)) if os.environ.get('INLINE', '0') == '1' else None
if flags is not None: L.mlsd_gemm_set_splitk_inline(1)    # ticket counters: slices added in the launch
VARIANTS = tuple(int(v) for v in os.environ.get('VARIANTS', '1,0').split(','))          # 1: 64x128x64s2, 0: 128x128x64s2, 23 / 5: their 3-deep rings
SHAPES = [(512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (512, 1280, 5120), (2048, 640, 2560), (128, 1280, 1280), (512, 3840, 1280),
          (2048, 1920, 640), (8192, 960, 320), (8192, 320, 1280), (128, 1280, 5120), (512, 1280, 11520), (128, 1280, 11520), (2048, 640, 5760), (8192, 320, 2880)]
for (M, N, K) in SHAPES:
    A = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16)); W = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    R = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)); C = _lib.DeviceBuffer(M * N * 4)
    res = []
    for v in VARIANTS:
        for ks in (1, 2, 3, 4, 6, 8, 12, 16):
            nkt = K // 64
            if ks > 1 and (nkt // ks < 2 or kernels.gemm_splitk_ws_bytes(M, N, ks) > (128 << 20)): continue
            a = kernels.GemmArgs(A=A.ptr, lda=K, W_=W.ptr, ldb=K, M=M, N=N, K=K, resid=R.ptr, ldr=N, C32=C.ptr, ldc32=N, tile_variant=v + 1, ksplit=ks,
                                 ws=ws.ptr, ws_bytes=128 << 20)
            if flags is not None: a.sk_flags = flags.ptr
            for _ in range(3): kernels.gemm(a)
            L.mlsd_event_record(ev[0], None)
            for _ in range(reps): kernels.gemm(a)
            L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
            ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
            res.append((ms.value / reps * 1e3, v, ks))
    best = {v: min(r for r in res if r[1] == v) for v in VARIANTS}
    line = " | ".join(f"v{v}: k/1 {[r[0] for r in res if r[1] == v and r[2] == 1][0]:6.1f}  best {best[v][0]:6.1f} (k/{best[v][2]})" for v in VARIANTS)
    print(f"{M:6d}x{N:5d}x{K:6d} f32+res  {line}")
