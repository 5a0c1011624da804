"""A linear launch that ENDS with the LayerNorm of its output (mlsd_gemm_args.ln_*) against the same launch followed by mlsd_layernorm; warm, back to back.
usage: python3 tools/ln_fold_bench.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps * 1e3
for (M, N, Kd) in [(8192, 1280, 1280), (8192, 1280, 5120)]:
    A = _lib.from_numpy(rng.standard_normal((M, Kd)).astype(np.float16)); W = _lib.from_numpy((rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16))
    R = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)); B = _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    G = _lib.from_numpy(np.ones(N, np.float32)); Bt = _lib.from_numpy(np.zeros(N, np.float32))
    C = _lib.DeviceBuffer(M * N * 4); Y = _lib.DeviceBuffer(M * N * 2)
    ws = _lib.from_numpy(np.zeros((M // 128) * (N // 320) * 512, np.uint32)); cnt = _lib.from_numpy(np.zeros(8192, np.uint32))
    a0 = kernels.GemmArgs(A=A.ptr, lda=Kd, W_=W.ptr, ldb=Kd, M=M, N=N, K=Kd, bias=B.ptr, C32=C.ptr, ldc32=N, resid=R.ptr, ldr=N, tile_variant=19)
    a1 = kernels.GemmArgs(A=A.ptr, lda=Kd, W_=W.ptr, ldb=Kd, M=M, N=N, K=Kd, bias=B.ptr, C32=C.ptr, ldc32=N, resid=R.ptr, ldr=N, tile_variant=19,
                          ln_y16=Y.ptr, ldln=N, ln_gamma=G.ptr, ln_beta=Bt.ptr, ln_eps=1e-5, ln_ws=ws.ptr, ln_cnt=cnt.ptr)
    t_g = timeit(lambda: kernels.gemm(a0)); t_l = timeit(lambda: kernels.layernorm(C.ptr, N, M, N, 1e-5, G.ptr, Bt.ptr, Y.ptr))
    t_gl = timeit(lambda: (kernels.gemm(a0), kernels.layernorm(C.ptr, N, M, N, 1e-5, G.ptr, Bt.ptr, Y.ptr)))
    t_f = timeit(lambda: kernels.gemm(a1))
    print(f"{M}x{N}x{Kd}: gemm {t_g:.1f} us, layernorm {t_l:.1f} us, both {t_gl:.1f} us, gemm ending with the layernorm {t_f:.1f} us", flush=True)
