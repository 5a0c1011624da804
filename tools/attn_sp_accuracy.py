"""Error of the attention kernels against a float64 softmax(QK^T / sqrt(d)) V on the same fp16 operands: the tile-loop kernels (fp32 scale behind the MFMA) and
attn64x2s (Q pre-scaled by log2(e) / sqrt(d) in fp16: one more rounding of q).  usage: python3 tools/attn_sp_accuracy.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib()
def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)
for (nb, heads, tq, tk, scale) in [(1, 4, 256, 256, 1.0), (1, 4, 1024, 1024, 1.0), (1, 4, 1024, 1024, 1.5), (1, 2, 1024, 4096, 1.0), (1, 2, 1024, 4096, 0.5)]:
    dh = 64; D = heads * dh
    rng = np.random.default_rng(tq + tk)
    q = (rng.standard_normal((nb, tq, D)) * scale).astype(np.float16); k = (rng.standard_normal((nb, tk, D)) * scale).astype(np.float16); v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    ref = np.empty((nb, tq, D))
    for b in range(nb):
        for h in range(heads):
            sl = slice(h * dh, (h + 1) * dh)
            s = q[b, :, sl].astype(np.float64) @ k[b, :, sl].astype(np.float64).T / 8.0
            p = np.exp(s - s.max(axis=1, keepdims=True)); p /= p.sum(axis=1, keepdims=True)
            ref[b, :, sl] = p @ v[b, :, sl].astype(np.float64)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D, bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
    res = {}
    for name, old, x2min, sp in (("32-row loop", 1, 2048, 0), ("64-row loop", 0, 256, 0), ("64-row pipelined", 0, 256, 1)):
        L.mlsd_attention_force_old(old); L.mlsd_attention_x2_min_tq(x2min); L.mlsd_attention_sp(2 if sp else 0)
        kernels.attention(a); res[name] = rel(do.download((nb, tq, D), np.float16).astype(np.float64), ref)
    print(f"h{heads} {tq}x{tk} operands x{scale}: rel-L2 against float64: " + "  ".join(f"{n} {e:.3e}" for n, e in res.items()), flush=True)
L.mlsd_attention_force_old(0); L.mlsd_attention_x2_min_tq(2048); L.mlsd_attention_sp(1)
