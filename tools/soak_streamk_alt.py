"""Stale-data screen of the stream-K hand-offs (tile variants 19 and 28): TWO operand sets with different values launched alternately over the SAME slab workspace and flags, every
fifth launch compared bit for bit with its set's first launch.  (A soak on fixed operands cannot see a stale slab: the previous launch's partial sums ARE the fresh ones -- the lesson
of round 5's LayerNorm-exchange hazard, profiles/NOTES.md.)  usage: python3 tools/soak_streamk_alt.py [launches]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
L = _lib.lib()
L.mlsd_gemm_streamk_ws_bytes.restype = ctypes.c_size_t
rng = np.random.default_rng(0)
ws = _lib.DeviceBuffer(L.mlsd_gemm_streamk_ws_bytes())
fl = _lib.from_numpy(np.zeros(4096, np.uint32))
bad = 0
t0 = time.time()
for (v, M, N, K) in [(28, 8192, 320, 2880), (28, 2048, 640, 5760), (28, 512, 1280, 11520), (28, 8192, 640, 5760), (19, 8192, 640, 5760), (19, 2048, 640, 17280), (19, 16384, 512, 16384), (28, 128, 1280, 11520)]:
    sets = []
    for k in range(2):
        sets.append(dict(A=_lib.from_numpy((rng.standard_normal((M, K)) * (1 + k)).astype(np.float16)), C=_lib.DeviceBuffer(M * N * 4)))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    args = [kernels.GemmArgs(A=d["A"].ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, C32=d["C"].ptr, ldc32=N, tile_variant=v + 1, ws=ws.ptr, ws_bytes=ws.nbytes, sk_flags=fl.ptr) for d in sets]
    name = kernels.gemm_variant(args[0])
    if "ppsk" not in name:
        print(f"{name} {M}x{N}x{K}: not a stream-K launch, skipped"); continue
    first = []
    for k in range(2):
        kernels.gemm(args[k]); first.append(sets[k]["C"].download((M * N,), np.uint32))
    bad0 = bad
    for r in range(reps):
        kernels.gemm(args[r & 1])
        if r % 10 >= 8 or r >= reps - 2:
            if not np.array_equal(sets[r & 1]["C"].download((M * N,), np.uint32), first[r & 1]):
                bad += 1; print("MISMATCH", name, M, N, K, "at launch", r, "operand set", r & 1, flush=True)
    if fl.download((4096,), np.uint32).any():
        bad += 1; print("FLAGS LEFT", name)
    print(f"{name} {M}x{N}x{K}: {reps} launches on two alternating operand sets, every fifth checked: {'MISMATCHES' if bad > bad0 else 'ok'}", flush=True)
print("soak", "FAILED" if bad else "passed", f"in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
