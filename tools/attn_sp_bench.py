"""Self attention at d_head 64 (SDXL shapes): the software-pipelined 64-rows-per-wave kernel (attn64x2s, round 6) against the tile-loop kernels; bit-identity against attn64x2.
usage: python3 tools/attn_sp_bench.py [reps]"""
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
    return ms.value / reps
rng = np.random.default_rng(0)
SHAPES = [(8, 10, 4096, 4096), (8, 20, 1024, 1024), (4, 10, 4096, 4096), (4, 20, 1024, 1024), (2, 20, 1024, 1024), (2, 10, 4096, 4096)]
for (nb, heads, tq, tk) in SHAPES[:int(os.environ.get('SP_SHAPES', '6'))]:
    dh = 64; D = heads * dh
    q = rng.standard_normal((nb, tq, D)).astype(np.float16); k = rng.standard_normal((nb, tk, D)).astype(np.float16); v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D, bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads,
                         d_head=dh, Tq=tq, Tk=tk, causal=0)
    outs = {}
    for name, old, x2min, sp in (("32 rows/wave tile loop", 1, 2048, 0), ("64 rows/wave tile loop", 0, 256, 0), ("64 rows/wave software-pipelined", 0, 256, 1)):
        L.mlsd_attention_force_old(old); L.mlsd_attention_x2_min_tq(x2min); L.mlsd_attention_sp(2 if sp else 0)
        ts = sorted(timeit(lambda: kernels.attention(a)) for _ in range(3))
        outs[name] = do.download((nb, tq, D), np.float16)
        print(f"attn b{nb} h{heads} {tq}x{tk} {name:34s}: {ts[0]*1e3:8.1f} us (median {ts[1]*1e3:8.1f})  {4.0*nb*heads*tq*tk*dh/ts[0]/1e9:7.1f} TFLOP/s", flush=True)
    same = np.array_equal(outs["64 rows/wave tile loop"].view(np.uint16), outs["64 rows/wave software-pipelined"].view(np.uint16))
    d = np.abs(outs["64 rows/wave software-pipelined"].astype(np.float32) - outs["32 rows/wave tile loop"].astype(np.float32)).max()
    print(f"   software-pipelined == 64-row tile loop bit for bit: {same};  max |diff| against the 32-row kernel {d:.2e}")
L.mlsd_attention_force_old(0); L.mlsd_attention_x2_min_tq(2048); L.mlsd_attention_sp(1)
