"""Phase stamps of the ping-pong attention kernel (block 0): where a tile's two phases spend their cycles.
usage: python3 tools/attn_trace.py [mode ...]     mode = mlsd_attention_pp value (4 = 32 rows 1 block/CU, 3 = 64 rows, 2 = 32 rows 2/CU)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
modes = [int(x) for x in sys.argv[1:]] or [4, 3, 2]
nb, heads, dh, tq, tk = 8, 10, 64, 4096, 4096
D = heads * dh
rng = np.random.default_rng(0)
q, k, v = (rng.standard_normal((nb, t, D)).astype(np.float16) for t in (tq, tk, tk))
dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
do = _lib.DeviceBuffer(nb * tq * D * 2)
a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D,
                     bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
tb = _lib.DeviceBuffer(8 * 16 * 5 * 8)
L.mlsd_attention_set_trace.argtypes = [vp]
for mode in modes:
    L.mlsd_attention_pp(mode)
    for _ in range(3): kernels.attention(a)
    L.mlsd_attention_set_trace(vp(tb.ptr))
    kernels.attention(a)
    L.mlsd_attention_set_trace(None)
    s = tb.download((8, 16, 5), np.uint64).astype(np.int64)
    print(f"mode {mode}: per tile (cycles): vector phase | wait at barrier | matrix phase | wait at barrier      [tiles 4..11]")
    for w in (0, 1, 4, 5):
        d = s[w, 4:12]
        vec = (d[:, 1] - d[:, 0]).mean(); b1 = (d[:, 2] - d[:, 1]).mean(); mat = (d[:, 3] - d[:, 2]).mean(); b2 = (d[:, 4] - d[:, 3]).mean()
        per = (s[w, 11, 0] - s[w, 4, 0]) / 7.0
        print(f"  wave {w}: vector {vec:7.0f} | {b1:6.0f} | matrix {mat:7.0f} | {b2:6.0f}   tile period {per:7.0f}")
L.mlsd_attention_pp(0)
