"""Race soak for the round-5 kernel: the 128x160 two-tiles-per-CU GEMM (tile variant 30) in every form that hands data over inside the launch -- the LayerNorm exchange with
2 / 4 / 8 partner tiles per row block, grids of up to 512 blocks (two per CU: all resident together), K = 128 .. 5120 -- many launches on TWO alternating
operand sets that share one scratch block, fp32 output and fp16 rows bit-identical to the set's first launch, counters back at zero; then the plans that use it (SD1.5 batch 1 as a hipGraph, SDXL batch 2
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
ws = _lib.from_numpy(np.zeros((2 << 20) // 4, np.uint32))           # ONE scratch block for every case, as in a plan: the partials of the previous launch (other shape, other values) lie where the next one reads
CASES = [(8192, 320, 320, 1), (8192, 320, 1280, 1), (2048, 640, 640, 1), (2048, 640, 2560, 0), (4096, 1280, 1280, 1), (4096, 1280, 5120, 1), (8192, 1280, 1280, 1),
         (16384, 640, 640, 1), (128, 160, 128, 0), (8192, 1280, 5120, 0)]
if os.environ.get("MLSD_TT_LN_ANYGRID") == "1":      # grids of more than two blocks per CU (partner tiles resident by dispatch order): not taken by the plan, soaked all the same
    CASES = [(32768, 640, 640, 1), (8192, 1280, 5120, 1), (32768, 640, 2560, 0)]
    ws = _lib.from_numpy(np.zeros((2 << 20) // 4, np.uint32))
VARIANT, TAG = 31, "128x160x64tt"
if len(sys.argv) > 3 and sys.argv[3] == "pp":       # the same alternating-operand soak on the ping-pong kernels' exchange (tile variant 18: single-round and whole-round grids)
    VARIANT, TAG = 19, "128x320x64pp"
    CASES = [(8192, 1280, 1280, 1), (8192, 1280, 5120, 1), (32768, 640, 640, 1), (32768, 640, 2560, 0), (4096, 1280, 1280, 1), (16384, 1280, 320, 1), (128, 1280, 192, 1)]
    ws = _lib.from_numpy(np.zeros((2 << 20) // 4, np.uint32))
for (M, N, K, res) in CASES:
    # TWO operand sets with different values, launched alternately: a tile that read a partner's partials of the PREVIOUS launch would produce other bits than its set's first launch
    # (with one set the stale values are the fresh ones: the first version of this soak could not see the bug it was written for)
    bad0 = bad
    ws = _lib.from_numpy(np.zeros((2 << 20) // 4, np.uint32))      # a region per op, as in a plan (a tile takes its tag from its own record of the op's previous launch); the two operand sets share it
    sets = []
    for k in range(2):
        sets.append(dict(A=_lib.from_numpy((rng.standard_normal((M, K)) * (1 + k)).astype(np.float16)), R=_lib.from_numpy((rng.standard_normal((M, N)) * 3 + 1 + 5 * k).astype(np.float32)),
                         C=_lib.DeviceBuffer(M * N * 4), Y=_lib.DeviceBuffer(M * N * 2)))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    dG, dB = _lib.from_numpy((1 + 0.2 * rng.standard_normal(N)).astype(np.float32)), _lib.from_numpy(rng.standard_normal(N).astype(np.float32))
    def mk(d):
        a = kernels.GemmArgs(A=d["A"].ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=d["C"].ptr, ldc32=N, tile_variant=VARIANT,
                             ln_y16=d["Y"].ptr, ldln=N, ln_gamma=dG.ptr, ln_beta=dB.ptr, ln_eps=1e-5, ln_ws=ws.ptr, ln_cnt=cnt.ptr)
        if res: a.resid, a.ldr = d["R"].ptr, N
        return a
    args = [mk(d) for d in sets]
    name = kernels.gemm_variant(args[0])
    assert TAG in name and "layernorm" in name, name
    get = lambda d: (d["C"].download((M * N,), np.uint32), d["Y"].download((M * N // 2,), np.uint32))
    first = []
    for k in range(2):
        kernels.gemm(args[k]); first.append(get(sets[k]))
    for r in range(reps):
        kernels.gemm(args[r & 1])
        if r % 10 >= 8 or r >= reps - 2:
            now = get(sets[r & 1])
            if not (np.array_equal(now[0], first[r & 1][0]) and np.array_equal(now[1], first[r & 1][1])):
                bad += 1; print("MISMATCH", name, M, N, K, "at launch", r, "operand set", r & 1)
    if cnt.download((8192,), np.uint32)[8191]:      # (only the sticky give-up word must stay clear)
        bad += 1; print("COUNTERS LEFT", name, M, N, K)
    print(f"{name} {M}x{N}x{K} ({N // (160 if VARIANT == 31 else 320)} partner tiles, {(M // 128) * (N // (160 if VARIANT == 31 else 320))} tiles): {reps} launches on two alternating operand sets, every fifth checked: {'MISMATCHES (above)' if bad > bad0 else 'ok'}", flush=True)
Lh = engine._proto2()
Lh.mlctx_handoff_retries.restype = ctypes.c_int
st = vp(); L.mlsd_stream_create(ctypes.byref(st))        # (a hipGraph plan captures its launches: not on the NULL stream)
for (model, lat, n, flags, label) in [("sd1", 64, 2, 8, "SD1.5 b1 (hipGraph)"), ("sdxl", 128, 4, 0, "SDXL b2"), ("sdxl", 128, 2, 0, "SDXL b1"), ("sdxl", 128, 8, 0, "SDXL b4")]:
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
