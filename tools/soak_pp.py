"""Race soak for the ping-pong GEMM tiles and the full SDXL plan: many launches on fixed operands, every output must be
bit-identical to the first (no atomics anywhere on the path).  usage: python3 tools/soak_pp.py [launches_per_case] [unet_evals]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mlimgsynth_amd import _lib, kernels, engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(0)
bad = 0
t0 = time.time()
cases = [(17, 8192, 10240, 1280, 5), (17, 4096, 3840, 320, 0), (17, 32768, 5120, 640, 0), (18, 8192, 1280, 1280, 1), (18, 8192, 1280, 5120, 1),
         (18, 32768, 640, 640, 1), (18, 131072, 320, 320, 0), (17, 2048, 2560, 2048, 0)]
for (pp, M, N, K, act) in cases:
    dA = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    nout = N // 2 if act == 5 else N
    dC = _lib.DeviceBuffer(M * nout * 2)
    dR = _lib.from_numpy(rng.standard_normal((M, nout)).astype(np.float32)) if act == 1 else None
    a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C16=dC.ptr, ldc16=nout, act=act, tile_variant=pp + 1,
                         resid=dR.ptr if dR else None, ldr=nout)
    assert "pp" in kernels.gemm_variant(a)
    kernels.gemm(a)
    first = dC.download((M, nout), np.float16).view(np.uint16).copy()
    for r in range(reps):
        kernels.gemm(a)
        if r % 50 == 49 or r == reps - 1:
            if not np.array_equal(dC.download((M, nout), np.float16).view(np.uint16), first):
                bad += 1
                print("MISMATCH", pp, M, N, K, "at launch", r)
    print(f"gemm pp{pp} {M}x{N}x{K} act{act}: {reps} launches ok" if not bad else "...")
# compiled-in epilogue bodies (round 2): fp32 + residual with column statistics (output AND statistics must repeat), fp16 fast path
for (pp, M, N, K) in [(18, 8192, 1280, 1280), (17, 8192, 3840, 1280), (18, 32768, 640, 2560)]:
    dA = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    dR = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    dC32, dC16, dS = _lib.DeviceBuffer(M * N * 4), _lib.DeviceBuffer(M * N * 2), _lib.DeviceBuffer(M // 64 * 2 * N * 4)
    for kind in ("f32+res+stats", "f16"):
        a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, tile_variant=pp + 1)
        if kind == "f16":
            a.C16, a.ldc16 = dC16.ptr, N
        else:
            a.C32, a.ldc32, a.resid, a.ldr, a.colstats = dC32.ptr, N, dR.ptr, N, dS.ptr
        kernels.gemm(a)
        get = (lambda: dC16.download((M, N), np.float16).view(np.uint16).copy()) if kind == "f16" else \
              (lambda: np.concatenate([dC32.download((M * N,), np.float32).view(np.uint32), dS.download((M // (64 if pp == 18 else 128) * 2 * N,), np.float32).view(np.uint32)]))
        first = get()
        for r in range(reps // 2):
            kernels.gemm(a)
            if r % 100 == 99 or r == reps // 2 - 1:
                if not np.array_equal(get(), first):
                    bad += 1; print("MISMATCH", kind, pp, M, N, K, "at launch", r)
        print(f"gemm pp{pp} {M}x{N}x{K} {kind}: {reps // 2} launches ok")
# conv case
n, h, w, cin, cout = 8, 32, 32, 1280, 1280
x = _lib.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float16))
wt = _lib.from_numpy((rng.standard_normal((cout, 9 * cin)) / np.sqrt(9 * cin)).astype(np.float16))
for pp in (17, 18):
    dC = _lib.DeviceBuffer(n * h * w * cout * 2)
    a = kernels.GemmArgs(A=x.ptr, lda=cin, conv=1, n_img=n, H=h, W=w, Cin=cin, OH=h, OW=w, KH=3, KW=3, stride=1, pad=1, W_=wt.ptr,
                         ldb=9 * cin, M=n * h * w, N=cout, K=9 * cin, C16=dC.ptr, ldc16=cout, tile_variant=pp + 1)
    assert "pp" in kernels.gemm_variant(a)
    kernels.gemm(a)
    first = dC.download((n * h * w, cout), np.float16).view(np.uint16).copy()
    for r in range(reps // 3):
        kernels.gemm(a)
    if not np.array_equal(dC.download((n * h * w, cout), np.float16).view(np.uint16), first):
        bad += 1; print("MISMATCH conv", pp)
    print(f"conv pp{pp} 8x32x32x1280->1280: {reps // 3} launches ok")
# full plan
un = engine.Unet("sdxl", 64, 64, 4)
P = un.P
xx = rng.standard_normal((4, 4, 64, 64)).astype(np.float32) * 3
cond = rng.standard_normal((4, 77, P.n_ctx)).astype(np.float32)
label = rng.standard_normal((4, P.ch_adm_in)).astype(np.float32)
sigma = np.array([7.0, 0.5, 2.0, 14.0], np.float32)
un.run(xx, cond, label, sigma)
first = un.run(xx, cond, label, sigma)
for e in range(evals):
    if not np.array_equal(un.run(xx, cond, label, sigma).view(np.uint32), first.view(np.uint32)):
        bad += 1; print("MISMATCH unet eval", e)
print(f"sdxl plan 64x64 N=4: {evals} evaluations ok" if not bad else "...")
# the headline plan (producer statistics, batched projections, compiled-in tile table): fewer evaluations, it is 8x the work
un.ctx.destroy()
un = engine.Unet("sdxl", 128, 128, 8)
xx = rng.standard_normal((8, 4, 128, 128)).astype(np.float32) * 3
cond = rng.standard_normal((8, 77, P.n_ctx)).astype(np.float32)
label = rng.standard_normal((8, P.ch_adm_in)).astype(np.float32)
sigma = np.array([7.0, 0.5, 2.0, 14.0, 1.0, 3.0, 0.1, 9.0], np.float32)
first = un.run(xx, cond, label, sigma)
for e in range(max(evals // 4, 10)):
    if not np.array_equal(un.run(xx, cond, label, sigma).view(np.uint32), first.view(np.uint32)):
        bad += 1; print("MISMATCH headline unet eval", e)
print(f"sdxl plan 128x128 N=8: {max(evals // 4, 10)} evaluations ok" if not bad else "...")
print("SOAK", "FAILED" if bad else "PASSED", f"({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
