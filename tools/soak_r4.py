"""Race soak for the round-4 kernels: stream-K on the 128x320 tile (batched gather), the skinny weight-streaming kernel, the split-K reduce pass that ends with the
LayerNorm, the one-dispatch GroupNorm -- many launches on fixed operands, every output bit-identical to the first -- then the SD1.5 batch-1 plan (hipGraph replay) and the
SDXL batch-4 plan with its weights streamed (uploads racing the launches of the previous segment), every evaluation bit-identical to the first.
usage: python3 tools/soak_r4.py [launches_per_case] [sd15_evals] [streamed_evals]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels, engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ev15 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
evst = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L = _lib.lib(); vp = _lib.vp
L.mlsd_gemm_streamk_ws_bytes.restype = ctypes.c_size_t
rng = np.random.default_rng(0)
bad = 0
t0 = time.time()
ws = _lib.DeviceBuffer(L.mlsd_gemm_streamk_ws_bytes())
fl = _lib.from_numpy(np.zeros(4096, np.uint32))
cases = [("sk320", 28, 8192, 320, 2880, 1), ("sk320", 28, 2048, 640, 5760, 1), ("sk320", 28, 512, 1280, 11520, 1), ("sk320", 28, 128, 1280, 11520, 1),
         ("skinny", 29, 128, 1280, 11520, 13), ("skinny", 29, 128, 1280, 1280, 10), ("skinny", 29, 2, 20160, 1280, 3), ("reduce+ln", 1, 512, 1280, 1280, 3),
         ("reduce+ln", 0, 2048, 640, 2560, 6)]
for (kind, v, M, N, K, ks) in cases:
    dA = _lib.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    dR = _lib.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    dC, dY = _lib.DeviceBuffer(M * N * 4), _lib.DeviceBuffer(M * N * 2)
    dG, dB = _lib.from_numpy(np.ones(N, np.float32)), _lib.from_numpy(np.zeros(N, np.float32))
    a = kernels.GemmArgs(A=dA.ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=dC.ptr, ldc32=N, resid=dR.ptr, ldr=N, tile_variant=v + 1, ksplit=ks, ws=ws.ptr, ws_bytes=ws.nbytes,
                         sk_flags=fl.ptr)
    if kind == "reduce+ln":
        a.sk_flags = None
        a.ln_y16, a.ldln, a.ln_gamma, a.ln_beta, a.ln_eps = dY.ptr, N, dG.ptr, dB.ptr, 1e-5
    name = kernels.gemm_variant(a)
    kernels.gemm(a)
    get = lambda: np.concatenate([dC.download((M * N,), np.uint32), dY.download((M * N // 2,), np.uint32) if kind == "reduce+ln" else np.zeros(0, np.uint32)])
    first = get()
    for r in range(reps):
        kernels.gemm(a)
        if r % 250 == 249 or r == reps - 1:
            if not np.array_equal(get(), first):
                bad += 1; print("MISMATCH", name, M, N, K, "at launch", r)
    if fl.download((4096,), np.uint32).any():
        bad += 1; print("FLAGS LEFT", name)
    print(f"{name} {M}x{N}x{K}: {reps} launches ok", flush=True)
# one-dispatch GroupNorm
n_img, hw, C = 2, 256, 1280
dX = _lib.from_numpy(rng.standard_normal((n_img * hw, C)).astype(np.float32))
dG, dB = _lib.from_numpy(np.ones(C, np.float32)), _lib.from_numpy(np.zeros(C, np.float32))
dY = _lib.DeviceBuffer(n_img * hw * C * 2)
L.mlsd_groupnorm_ws_bytes.restype = ctypes.c_size_t
gws = _lib.DeviceBuffer(L.mlsd_groupnorm_ws_bytes(n_img, hw, 32))
g = kernels.GnArgs(x1=dX.ptr, ld1=C, C1=C, C2=0, n_img=n_img, HW=hw, n_grp=32, eps=1e-6, gamma=dG.ptr, beta=dB.ptr, silu=1, y16=dY.ptr, ws=gws.ptr)
kernels.groupnorm(g)
first = dY.download((n_img * hw * C,), np.uint16)
for r in range(reps):
    kernels.groupnorm(g)
if not np.array_equal(dY.download((n_img * hw * C,), np.uint16), first):
    bad += 1; print("MISMATCH groupnorm one dispatch")
print(f"groupnorm n{n_img} hw{hw} c{C} one dispatch: {reps} launches ok", flush=True)

# SD1.5 batch-1 plan as a hipGraph
st = vp(); _lib.check(L.mlsd_stream_create(ctypes.byref(st)), "stream")
un = engine.Unet("sd1", 64, 64, 2, stream=st.value, flags=8)
P = un.P
x = rng.standard_normal((2, 4, 64, 64)).astype(np.float32) * 3
cond = rng.standard_normal((2, 77, P.n_ctx)).astype(np.float32)
sig = np.array([7.0, 0.5], np.float32)
first = un.run(x, cond, None, sig).view(np.uint32).copy()
for r in range(ev15):
    if not np.array_equal(un.run(x, cond, None, sig).view(np.uint32), first):
        bad += 1; print("MISMATCH sd15 plan at evaluation", r); break
print(f"sd15 b1 plan (hipGraph): {ev15} evaluations ok", flush=True)
un.ctx.destroy()
# SDXL batch-4 plan, weights streamed through two 512 MiB slabs, against itself and against the resident plan
res = engine.Unet("sdxl", 128, 128, 8)
P = res.P
x = rng.standard_normal((8, 4, 128, 128)).astype(np.float32) * 3
cond = rng.standard_normal((8, 77, P.n_ctx)).astype(np.float32)
lab = rng.standard_normal((8, P.ch_adm_in)).astype(np.float32)
sig = np.linspace(9, 0.3, 8).astype(np.float32)
want = res.run(x, cond, lab, sig).view(np.uint32).copy()
res.ctx.destroy()
s = engine.Unet("sdxl", 128, 128, 8, stream_weights_mib=512)
for r in range(evst):
    if not np.array_equal(s.run(x, cond, lab, sig).view(np.uint32), want):
        bad += 1; print("MISMATCH streamed sdxl plan at evaluation", r); break
print(f"sdxl b4 plan, weights streamed: {evst} evaluations equal to the resident plan", flush=True)
print(f"soak done in {time.time() - t0:.0f} s: {'ALL BIT-IDENTICAL' if not bad else str(bad) + ' MISMATCHES'}")
sys.exit(1 if bad else 0)
