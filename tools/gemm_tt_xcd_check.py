"""Where do the partner tiles of a row block land?  The 128x160 kernel's per-block stamps carry HW_ID and XCC_ID: for grids of 256 / 512 / 1024 blocks, how many row blocks have all of their
N / 160 tiles on ONE XCD (what the LayerNorm exchange through that XCD's L2 needs)?  usage: python3 tools/gemm_tt_xcd_check.py"""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels
L = _lib.lib()
rng = np.random.default_rng(0)
for (M, N, Kd) in [(4096, 1280, 1280), (8192, 640, 640), (8192, 1280, 1280), (8192, 1280, 5120), (16384, 640, 640), (32768, 640, 640)]:
    A = _lib.from_numpy(rng.standard_normal((M, Kd)).astype(np.float16)); W = _lib.from_numpy((rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16))
    C = _lib.DeviceBuffer(M * N * 4)
    nbm, nbn = M // 128, N // 160
    nblk = nbm * nbn
    T = _lib.from_numpy(np.zeros((nblk, 4), np.uint64))
    a = kernels.GemmArgs(A=A.ptr, lda=Kd, W_=W.ptr, ldb=Kd, M=M, N=N, K=Kd, C32=C.ptr, ldc32=N, tile_variant=31)
    split_total, launches = 0, 20
    worst = 0
    for it in range(launches):
        L.mlsd_gemm_tt_set_trace(ctypes.c_void_p(T.ptr)); kernels.gemm(a); kernels.sync(); L.mlsd_gemm_tt_set_trace(None)
        t = T.download((nblk, 4), np.uint64)
        xcc = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
        v = np.arange(nblk)
        q8, x8, j8 = nblk >> 3, v & 7, v >> 3
        bid = x8 * q8 + j8                      # (nblk is a multiple of 8 here)
        bmi = bid // nbn
        split = sum(1 for r in range(nbm) if len(set(xcc[bmi == r].tolist())) > 1)
        mism = int(np.sum(xcc != (v & 7)))     # blocks NOT on XCD id % 8
        split_total += split; worst = max(worst, mism)
    print(f"{M}x{N}x{Kd}: {nblk} blocks ({nbn} partner tiles): row blocks whose tiles sit on more than one XCD: {split_total} in {launches} launches ({nbm} row blocks each); "
          f"blocks not on XCD (id % 8): up to {worst} per launch", flush=True)
