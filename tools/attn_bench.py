"""Attention micro-benchmark on the SDXL self-attention shapes (diagnostics / PMC target)."""
import ctypes, sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
HAS_EXP = bool(L.mlsd_has_experiments())
print('library built with EXPERIMENTS:', HAS_EXP, flush=True)
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)
def timeit(fn):
    for _ in range(3): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps

# ---- the 77-key cross attention: general tile-loop kernels against the one-pass kernel (query blocks per workgroup, register budget)
for (nb, heads, dh, tq, tk) in [(8, 20, 64, 1024, 77), (8, 10, 64, 4096, 77), (2, 8, 40, 4096, 77), (2, 8, 160, 256, 77)]:
    D = heads * dh
    q = rng.standard_normal((nb, tq, D)).astype(np.float16); k = rng.standard_normal((nb, tk, D)).astype(np.float16)
    v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D,
                         bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
    ref = None
    for name, on, qb in [("general kernels", 0, 0), ("one pass, auto", 1, 0), ("one pass, qb 1", 1, 1), ("one pass, qb 2", 1, 2), ("one pass, qb 4", 1, 4),
                         ("one pass, qb 8", 1, 8), ("one pass w4, auto", 2, 0), ("one pass w4, qb 1", 2, 1), ("one pass w4, qb 2", 2, 2)]:
        L.mlsd_attention_tk96(on, qb)
        t = timeit(lambda: kernels.attention(a))
        o = do.download((nb, tq, D), np.float16).astype(np.float32)
        if ref is None: ref = o
        gb = 2.0 * nb * D * (2 * tq + 2 * tk) / 1e9
        print(f"attn b{nb} h{heads} d{dh} {tq}x{tk} {name:18s}: {t*1e3:8.1f} us  {4.0*nb*heads*tq*tk*dh/t/1e9:7.1f} TFLOP/s  {gb/t/1e3:6.2f} TB/s   max|diff vs general| {np.abs(o - ref).max():.2e}", flush=True)
    L.mlsd_attention_tk96(1, 0)

# ---- self attention of the SDXL UNet: tile-loop kernels against the ping-pong kernel builds
for (nb, heads, dh, tq, tk) in [(8, 10, 64, 4096, 4096), (8, 20, 64, 1024, 1024), (4, 10, 64, 4096, 4096), (4, 20, 64, 1024, 1024)]:
    D = heads * dh
    q = rng.standard_normal((nb, tq, D)).astype(np.float16); k = rng.standard_normal((nb, tk, D)).astype(np.float16)
    v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D,
                         bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
    L.mlsd_attention_x2_min_tq(256)                      # let the 64-rows/wave kernel take every shape here
    ref = None
    for name, pp, old in (("general (32 rows/wave)", 0, 1), ("64 rows/wave, LDS-DMA ", 0, 0), ("ping-pong 32 rows, 2/CU", 2, 0),
                          ("ping-pong 32 rows, 1/CU", 4, 0), ("ping-pong 64 rows      ", 3, 0),
                          ("pp 64 rows, prio M    ", 3 + 16, 0), ("pp 64 rows, prio V    ", 3 + 32, 0),
                          ("pp 64 rows NO VECTOR PHASE (timing only)", 3 + 256, 0), ("pp 64 rows NO MATRIX PHASE (timing only)", 3 + 512, 0),
                          ("pp 32 rows 1/CU NO VECTOR PHASE (timing only)", 4 + 256, 0), ("pp 32 rows 1/CU NO MATRIX PHASE (timing only)", 4 + 512, 0)):
        if pp and not HAS_EXP: continue       # the ping-pong attention builds exist only with `make EXPERIMENTS=1`: the product build would silently time the 64-row kernel under their names
        L.mlsd_attention_pp(pp); L.mlsd_attention_force_old(old); L.mlsd_attention_vsum(1)
        ts = sorted(timeit(lambda: kernels.attention(a)) for _ in range(3))
        o = do.download((nb, tq, D), np.float16).astype(np.float32)
        if ref is None: ref = o
        print(f"attn b{nb} h{heads} d{dh} {tq}x{tk} {name}: {ts[0]*1e3:8.1f} us (median {ts[1]*1e3:8.1f})  {4.0*nb*heads*tq*tk*dh/ts[0]/1e9:7.1f} TFLOP/s   max|diff vs general| {np.abs(o - ref).max():.2e}", flush=True)
L.mlsd_attention_force_old(0); L.mlsd_attention_vsum(1); L.mlsd_attention_x2_min_tq(2048); L.mlsd_attention_pp(0)
