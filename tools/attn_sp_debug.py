import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib()
for (nb, heads, tq, tk, scale) in [(2, 10, 1024, 1024, 1.5), (8, 10, 4096, 4096, 1.0)]:
    dh = 64; D = heads * dh
    rng = np.random.default_rng(tq + tk)
    q = (rng.standard_normal((nb, tq, D)) * scale).astype(np.float16); k = (rng.standard_normal((nb, tk, D)) * scale).astype(np.float16); v = rng.standard_normal((nb, tk, D)).astype(np.float16)
    dq, dk, dv = _lib.from_numpy(q), _lib.from_numpy(k), _lib.from_numpy(v)
    do = _lib.DeviceBuffer(nb * tq * D * 2)
    a = kernels.AttnArgs(q=dq.ptr, k=dk.ptr, v=dv.ptr, out=do.ptr, ldq=D, ldk=D, ldv=D, ldo=D, bsq=tq * D, bsk=tk * D, bsv=tk * D, bso=tq * D, n_batch=nb, n_head=heads, d_head=dh, Tq=tq, Tk=tk, causal=0)
    L.mlsd_attention_x2_min_tq(256)
    L.mlsd_attention_sp(0); kernels.attention(a); loop = do.download((nb, tq, D), np.float16).astype(np.float32)
    L.mlsd_attention_sp(2)
    runs = []
    for i in range(6):
        kernels.attention(a); runs.append(do.download((nb, tq, D), np.float16).astype(np.float32))
    print(f"b{nb} h{heads} {tq}x{tk} scale {scale}: max |sp - loop| {np.abs(runs[0] - loop).max():.3e}")
    for i in range(1, 6):
        d = np.abs(runs[i] - runs[0])
        idx = np.argwhere(d > 0)
        print(f"   run {i} vs run 0: {len(idx)} elements differ, max {d.max():.3e}", end="")
        if len(idx):
            bs = sorted(set((int(b), int(r) // 256, int(c) // 64) for b, r, c in idx[:2000]))
            rows = sorted(set(int(r) % 64 for b, r, c in idx[:2000]))
            print(f"; (batch, 256-row block, head) {bs[:8]}; rows mod 64: {rows[:40]}; cols {sorted(set(int(c) % 64 for b, r, c in idx[:2000]))[:20]}", end="")
        print()
