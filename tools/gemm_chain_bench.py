"""Output projection + LayerNorm followed by the Linear that consumes it (the self-attention out projection and the cross-attention q projection of a transformer block):
as the plan ran them until round 5 (128x320 ping-pong tile ending with the LayerNorm, then the 128x160 kernel), both on the 128x160 kernel, and as ONE launch
(mlsd_gemm_args.chain_W).  usage: python3 tools/gemm_chain_bench.py [reps]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels  # noqa: E402

L = _lib.lib()
vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
ev = [vp(), vp()]
for e in ev:
    L.mlsd_event_create(ctypes.byref(e))


def timed(fn):
    best = 1e9
    for _ in range(3):
        fn(); fn()
        L.mlsd_event_record(ev[0], None)
        for _ in range(reps):
            fn()
        L.mlsd_event_record(ev[1], None)
        L.mlsd_event_sync(ev[1])
        ms = ctypes.c_float()
        L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
        best = min(best, ms.value / reps)
    return best * 1e3


for M, N in [(8192, 1280), (4096, 1280), (16384, 640), (8192, 320), (2048, 640)]:
    Kd = N
    ws = _lib.DeviceBuffer(4 << 20); cnt = _lib.from_numpy(np.zeros(8192, np.uint32))
    dW = _lib.from_numpy((rng.standard_normal((N, Kd)) / np.sqrt(Kd)).astype(np.float16)); dW2 = _lib.from_numpy((rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16))
    dB = _lib.from_numpy(rng.standard_normal(N).astype(np.float32)); dG = _lib.from_numpy(np.ones(N, np.float32))
    sets = []
    for k in range(2):      # two operand sets, alternated: nothing stays warmer than in the plan's layer-to-layer walk
        sets.append(dict(A=_lib.from_numpy(rng.standard_normal((M, Kd)).astype(np.float16)), R=_lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32)),
                         C=_lib.DeviceBuffer(M * N * 4), Y=_lib.DeviceBuffer(M * N * 2), Q=_lib.DeviceBuffer(M * N * 2)))

    def a_ln(d, variant, chain=False):
        a = kernels.GemmArgs(A=d["A"].ptr, lda=Kd, W_=dW.ptr, ldb=Kd, M=M, N=N, K=Kd, bias=dB.ptr, resid=d["R"].ptr, ldr=N, C32=d["C"].ptr, ldc32=N, tile_variant=variant + 1,
                             ln_y16=d["Y"].ptr, ldln=N, ln_gamma=dG.ptr, ln_beta=dB.ptr, ln_eps=1e-5, ln_ws=ws.ptr, ln_cnt=cnt.ptr)
        if chain:
            a.chain_W, a.chain_ldb, a.chain_bias, a.chain_C16, a.chain_ldc16 = dW2.ptr, N, dB.ptr, d["Q"].ptr, N
        return a

    def a_q(d, variant):
        return kernels.GemmArgs(A=d["Y"].ptr, lda=N, W_=dW2.ptr, ldb=N, M=M, N=N, K=N, bias=dB.ptr, C16=d["Q"].ptr, ldc16=N, tile_variant=variant + 1)
    out = []
    for name, v1, v2, chain in (("ping-pong + LN | 128x160", 18, 30, False), ("128x160 + LN | 128x160", 30, 30, False), ("ONE launch", 30, 30, True)):
        A1 = [a_ln(d, v1, chain) for d in sets]; A2 = [a_q(d, v2) for d in sets]
        if "layernorm" not in kernels.gemm_variant(A1[0]):
            out.append(f"{name}: n/a"); continue
        it = [0]

        def fn():
            k = it[0] & 1; it[0] += 1
            kernels.gemm(A1[k])
            if not chain:
                kernels.gemm(A2[k])
        out.append(f"{name}: {timed(fn):6.1f} us")
    print(f"{M}x{N}x{Kd} + {M}x{N}x{N}: " + " | ".join(out), flush=True)
