"""The q projection of a cross attention that ends with that attention (gemm_pp.hpp PP_EPI_XATTN) against the unfused pair, on the SDXL shapes; with the in-kernel stamps of the
fused launch (median over blocks): K loop, ring drain, q exchange + K landing, scores, P.V + stores.
usage: python3 tools/xattn_bench.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def timeit(fn):
    for _ in range(3): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps * 1e3
L.mlsd_gemm_set_xattn(2)      # also the launches the plan leaves unfused (fewer tiles than half the CUs)
L.mlsd_xattn_pack_vt.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
rng = np.random.default_rng(0)
for (nb, tq, D, kd) in [(8, 1024, 1280, 1280), (8, 4096, 640, 640), (4, 1024, 1280, 1280), (2, 1024, 1280, 1280)]:
    M, tk, heads = nb * tq, 77, D // 64
    nset = 6                                           # rotate operand sets: activations from HBM / MALL as in the plan, not from L2
    xs = [_lib.from_numpy(rng.standard_normal((M, kd)).astype(np.float16)) for _ in range(nset)]
    w = _lib.from_numpy((rng.standard_normal((D, kd)) / np.sqrt(kd)).astype(np.float16))
    ldkv = 2 * D
    kv = _lib.from_numpy(rng.standard_normal((nb * tk, ldkv)).astype(np.float16))
    vt = _lib.DeviceBuffer(nb * D * 96 * 2)
    _lib.check(L.mlsd_xattn_pack_vt(vp(kv.ptr + 2 * D), ldkv, nb, tk, D, vp(vt.ptr), None), "pack")
    outs = [_lib.DeviceBuffer(M * D * 2) for _ in range(nset)]
    qb = _lib.DeviceBuffer(M * D * 2)
    it = [0]
    def fused():
        i = it[0] % nset; it[0] += 1
        a = kernels.GemmArgs(A=xs[i].ptr, lda=kd, W_=w.ptr, ldb=kd, M=M, N=D, K=kd, xa_k=kv.ptr, xa_ldk=ldkv, xa_vt=vt.ptr, xa_out=outs[i].ptr, xa_ldo=D, xa_Tq=tq, xa_Tk=tk)
        kernels.gemm(a)
    def pair(variant):
        def run():
            i = it[0] % nset; it[0] += 1
            g = kernels.GemmArgs(A=xs[i].ptr, lda=kd, W_=w.ptr, ldb=kd, M=M, N=D, K=kd, C16=qb.ptr, ldc16=D, tile_variant=variant)
            kernels.gemm(g)
            at = kernels.AttnArgs(q=qb.ptr, k=kv.ptr, v=kv.ptr + 2 * D, out=outs[i].ptr, ldq=D, ldk=ldkv, ldv=ldkv, ldo=D, bsq=tq * D, bsk=tk * ldkv, bsv=tk * ldkv, bso=tq * D,
                                  n_batch=nb, n_head=heads, d_head=64, Tq=tq, Tk=tk, causal=0)
            kernels.attention(at)
        return run
    def proj(variant):
        def run():
            i = it[0] % nset; it[0] += 1
            g = kernels.GemmArgs(A=xs[i].ptr, lda=kd, W_=w.ptr, ldb=kd, M=M, N=D, K=kd, C16=qb.ptr, ldc16=D, tile_variant=variant)
            kernels.gemm(g)
        return run
    print(f"q projection {M}x{D}x{kd} + cross attention ({nb} images, {tq} rows, {heads} heads, {tk} keys):")
    print(f"   fused launch                               {timeit(fused):7.1f} us")
    print(f"   projection (128x160 two-per-CU) + attention {timeit(pair(31)):7.1f} us   (projection alone {timeit(proj(31)):6.1f})")
    print(f"   projection (128x320 ping-pong) + attention  {timeit(pair(19)):7.1f} us   (projection alone {timeit(proj(19)):6.1f})")
    tb = _lib.DeviceBuffer(2 * 32768 * 8)
    for _ in range(3): fused()
    L.mlsd_gemm_set_trace(vp(tb.ptr)); fused(); L.mlsd_device_sync(); L.mlsd_gemm_set_trace(None)
    nblk = min((M // 128) * (D // 320), 4096)
    raw = tb.download((2, 4096, 8), np.uint64).astype(np.int64)
    t, x = raw[0][:nblk], raw[1][:nblk]
    med = lambda v: float(np.median(v))
    life = med(t[:, 6] - t[:, 0])
    print(f"   in-kernel, shader clocks, median of {nblk} blocks (block life {life:.0f}):")
    for name, d in [("prologue", t[:, 1] - t[:, 0]), ("K loop", t[:, 2] - t[:, 1]), ("tail stages drained + barrier", x[:, 1] - x[:, 0]),
                    ("K DMA issued, q -> fp16 -> LDS, barrier", x[:, 2] - x[:, 1]), ("q fragments to registers, barrier", x[:, 3] - x[:, 2]),
                    ("V^T DMA issued, K landed, barrier", x[:, 4] - x[:, 3]), ("scores + softmax (5 heads)", x[:, 5] - x[:, 4]), ("V^T landed, barrier", x[:, 6] - x[:, 5]),
                    ("P.V, normalise, stores issued", x[:, 7] - x[:, 6]), ("store drain -> exit", t[:, 6] - x[:, 7])]:
        print(f"      {name:48s} {med(d):8.0f}  ({100 * med(d) / life:4.1f} %)")
