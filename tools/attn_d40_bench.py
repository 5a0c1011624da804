"""SD1.5's 4096-token self attention (b2 h8 d40): the general 32-row kernel against attn64x2s run as d = 64 with zero Q columns.  usage: python3 tools/attn_d40_bench.py [reps]"""
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
for (nb, heads, tq, tk) in [(2, 8, 4096, 4096), (4, 8, 4096, 4096), (8, 8, 4096, 4096)]:
    dh = 40; D = heads * dh
    qkv = rng.standard_normal((nb, tq, 3 * D)).astype(np.float16)
    d = _lib.from_numpy(qkv)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=d.ptr, k=d.ptr + 2 * D, v=d.ptr + 4 * D, out=do.ptr, ldq=3 * D, ldk=3 * D, ldv=3 * D, ldo=D, bsq=tq * 3 * D, bsk=tq * 3 * D, bsv=tq * 3 * D, bso=tq * D,
                         n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
    outs = {}
    for name, sp in (("general kernel", 0), ("pipelined as d 64", 1)):
        L.mlsd_attention_sp(sp)
        t = min(timeit(lambda: kernels.attention(a)) for _ in range(3))
        outs[name] = do.download((nb, tq, D), np.float16).astype(np.float32)
        print(f"attn b{nb} h{heads} d40 {tq}x{tk} {name:20s}: {t:7.1f} us  {4.0 * nb * heads * tq * tk * dh / t / 1e6:7.1f} TFLOP/s", flush=True)
    print(f"   max |diff| {np.abs(outs['general kernel'] - outs['pipelined as d 64']).max():.2e}")
L.mlsd_attention_sp(1)
