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
print("SOAK", "FAILED" if bad else "PASSED", f"({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
