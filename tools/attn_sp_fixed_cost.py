"""Fixed cost per launch / block of the d = 64 self-attention kernels: Tq = 1024, b8 h20 (SDXL's 1024-token level), the key count swept -- time = T0 + tiles x t.
usage: python3 tools/attn_sp_fixed_cost.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def timeit(fn):
    for _ in range(3): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps * 1e3
rng = np.random.default_rng(0)
nb, heads, dh, tq = 8, 20, 64, 1024
D = heads * dh
for sp in (2, 0):
    L.mlsd_attention_sp(sp); L.mlsd_attention_x2_min_tq(256 if sp == 0 else 2048)
    pts = []
    for tk in (128, 256, 512, 1024, 2048, 4096):
        q = rng.standard_normal((nb, tq, D)).astype(np.float16); k = rng.standard_normal((nb, tk, D)).astype(np.float16); v = rng.standard_normal((nb, tk, D)).astype(np.float16)
        dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
        do = _lib.DeviceBuffer(nb * tq * D * 2)
        a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D, bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
        t = min(timeit(lambda: kernels.attention(a)) for _ in range(3))
        pts.append((tk // 64, t))
        print(f"{'pipelined' if sp else '64-row tile loop'} b8 h20 1024 x {tk:5d}: {t:7.1f} us ({tk // 64:3d} key tiles)", flush=True)
    x = np.array([p[0] for p in pts], float); y = np.array([p[1] for p in pts], float)
    A = np.vstack([np.ones_like(x), x]).T
    (t0, tt), *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"   fit: {t0:.1f} us fixed + {tt:.2f} us per key tile (640 blocks of 256 rows on 512 slots)")
L.mlsd_attention_sp(1); L.mlsd_attention_x2_min_tq(2048)
