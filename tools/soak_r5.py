"""Race soak for the round-5 kernel: the 128x160 two-tiles-per-CU GEMM (tile variant 30) in every form that hands data over inside the launch -- the LayerNorm exchange with
2 / 4 / 8 partner tiles per row block, single-round and multi-round grids (1024 blocks on 512 slots: partner tiles resident by dispatch order), K = 128 .. 5120 -- many
launches on fixed operands, fp32 output and fp16 rows bit-identical to the first, counters back at zero; then the plans that use it (SD1.5 batch 1 as a hipGraph, SDXL batch 2
and batch 1), every evaluation bit-identical to the first and no hand-off retry.
usage: python3 tools/soak_r5.py [launches_per_case] [evals_per_plan]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels, engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 100
L = _lib.lib(); vp = _lib.vp
rng = np.random.default_rng(0)
bad = 0
t0 = time.time()
cnt = _lib.from_numpy(np.zeros(8192, np.uint32))
for (M, N, K, res) in [(8192, 320, 320, 1), (8192, 320, 1280, 1), (2048, 640, 640, 1), (2048, 640, 2560, 0), (4096, 1280, 1280, 1), (4096, 1280, 5120, 1), (8192, 1280, 1280, 1),
                       (32768, 640, 640, 1), (128, 160, 128, 0), (16384, 1280, 1280, 1)]:
    dA = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    dR = _lib.from_numpy((rng.standard_normal((M, N)) * 3 + 1).astype(np.float32))
    dC, dY = _lib.DeviceBuffer(M * N * 4), _lib.DeviceBuffer(M * N * 2)
    dG, dB = _lib.from_numpy((1 + 0.2 * rng.standard_normal(N)).astype(np.float32)), _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    ws = _lib.DeviceBuffer((M // 128) * (N // 160) * 1024)
    a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=dC.ptr, ldc32=N, tile_variant=31,
                         ln_y16=dY.ptr, ldln=N, ln_gamma=dG.ptr, ln_beta=dB.ptr, ln_eps=1e-5, ln_ws=ws.ptr, ln_cnt=cnt.ptr)
    if res: a.resid, a.ldr = dR.ptr, N
    name = kernels.gemm_variant(a)
    assert "128x160x64tt" in name and "layernorm" in name, name
    kernels.gemm(a)
    get = lambda: (dC.download((M * N,), np.uint32), dY.download((M * N // 2,), np.uint32))
    first = get()
    for r in range(reps):
        kernels.gemm(a)
        if r % 500 == 499 or r == reps - 1:
            now = get()
            if not (np.array_equal(now[0], first[0]) and np.array_equal(now[1], first[1])):
                bad += 1; print("MISMATCH", name, M, N, K, "at launch", r)
    if cnt.download((8192,), np.uint32).any():
        bad += 1; print("COUNTERS LEFT", name, M, N, K)
    print(f"{name} {M}x{N}x{K} ({N // 160} partner tiles, {(M // 128) * (N // 160)} blocks): {reps} launches ok", flush=True)
Lh = engine._proto2()
Lh.mlctx_handoff_retries.restype = ctypes.c_int
st = vp(); L.mlsd_stream_create(ctypes.byref(st))        # (a hipGraph plan captures its launches: not on the NULL stream)
for (model, lat, n, flags, label) in [("sd1", 64, 2, 8, "SD1.5 b1 (hipGraph)"), ("sdxl", 128, 4, 0, "SDXL b2"), ("sdxl", 128, 2, 0, "SDXL b1")]:
    un = engine.Unet(model, lat, lat, n, flags=flags, stream=st.value)
    ntt = sum(1 for l, _ in un.ctx.op_list() if "128x160x64tt" in l)
    P = un.P
    x = rng.standard_normal((n, 4, lat, lat)).astype(np.float32) * 3
    cond = rng.standard_normal((n, 77, P.n_ctx)).astype(np.float32)
    label_ = rng.standard_normal((n, P.ch_adm_in)).astype(np.float32) if P.ch_adm_in else None
    sigma = rng.uniform(0.5, 10.0, n).astype(np.float32)
    first = un.run(x, cond, label_, sigma).view(np.uint32).copy()
    for r in range(nev):
        if not np.array_equal(un.run(x, cond, label_, sigma).view(np.uint32), first):
            bad += 1; print("MISMATCH", label, "evaluation", r)
    print(f"{label}: {ntt} launches of the 128x160 kernel per evaluation, {nev} evaluations bit-identical", flush=True)
    un.ctx.destroy()
print("hand-off retries:", Lh.mlctx_handoff_retries())
print("soak", "FAILED" if bad or Lh.mlctx_handoff_retries() else "passed", f"in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
