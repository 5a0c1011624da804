"""Per-block timeline of the two-tiles-per-CU GEMM (variant 30): which blocks share a CU, when each block's K loop ends and when it exits.
usage: python3 tools/gemm_tt_trace.py [M N K] [prio]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib()
M, N, Kd = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 1280, 1280)
prio = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(0)
A = _lib.from_numpy(rng.standard_normal((M, Kd)).astype(np.float16)); W = _lib.from_numpy((rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16))
R = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)); C = _lib.DeviceBuffer(M * N * 4)
nblk = (M // 128) * (N // 160)
T = _lib.from_numpy(np.zeros((nblk, 4), np.uint64))
a = kernels.GemmArgs(A=A.ptr, lda=Kd, W_=W.ptr, ldb=Kd, M=M, N=N, K=Kd, C32=C.ptr, ldc32=N, resid=R.ptr, ldr=N, tile_variant=31)
L.mlsd_gemm_tt_set_prio(prio)
for _ in range(3): kernels.gemm(a)
kernels.sync()
L.mlsd_gemm_tt_set_trace(ctypes.c_void_p(T.ptr)); kernels.gemm(a); kernels.sync(); L.mlsd_gemm_tt_set_trace(None)
t = T.download((nblk, 4), np.uint64)
t0 = t[:, 0].min()
st, le, ex = [(t[:, i] - t0).astype(np.float64) / 100.0 for i in range(3)]       # us (100 MHz)
hw = t[:, 3]; cu = ((hw >> 32) & 0xf) * 4096 + (hw & 0xffffff00 & 0xff00) + 0     # (xcc, se/sh/cu bits 8..15)
print(f"{M}x{N}x{Kd} prio={prio}: {nblk} blocks; launch span {ex.max():.1f} us; distinct (xcc, cu) ids {len(set(cu.tolist()))}")
cls = (np.arange(nblk) >> 8) & 1
for c in (0, 1):
    m = cls == c
    if m.any(): print(f"  class {c} ({'high' if c == 0 else 'low'} priority): start {st[m].mean():6.2f} (max {st[m].max():6.2f})  loop end {le[m].mean():6.2f} (min {le[m].min():6.2f} max {le[m].max():6.2f})  exit {ex[m].mean():6.2f} (max {ex[m].max():6.2f})")
per = {}
for b in range(nblk): per.setdefault(int(cu[b]), []).append(b)
sizes = np.bincount([len(v) for v in per.values()])
print("  blocks per CU histogram:", {i: int(n) for i, n in enumerate(sizes) if n})
mixed = sum(1 for v in per.values() if len(set(int(cls[b]) for b in v)) > 1)
print(f"  CUs holding both classes: {mixed} of {len(per)}")
for k in list(per)[:6]:
    print("   cu", hex(k), [(b, int(cls[b]), round(st[b], 1), round(le[b], 1), round(ex[b], 1)) for b in per[k]])
