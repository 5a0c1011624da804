import ctypes, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def timeit(fn, reps=50):
    for _ in range(3): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps * 1e3
rng = np.random.default_rng(0)
for (nb, heads, tq, tk) in [(8, 20, 256, 256), (8, 20, 512, 512), (8, 10, 1024, 1024), (2, 20, 256, 256), (2, 10, 1024, 1024), (8, 10, 2304, 2304), (8, 20, 768, 768)]:
    dh = 64; D = heads * dh
    q = rng.standard_normal((nb, tq, D)).astype(np.float16); k = rng.standard_normal((nb, tk, D)).astype(np.float16); v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D, bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
    res = []
    for sp in (0, 2):
        L.mlsd_attention_sp(sp)
        res.append(min(timeit(lambda: kernels.attention(a)) for _ in range(3)))
    print(f"b{nb} h{heads} {tq}x{tk}: plan's kernel without sp {res[0]:7.1f} us | pipelined {res[1]:7.1f} us  ({res[1]/res[0]:.3f})", flush=True)
L.mlsd_attention_sp(1)
