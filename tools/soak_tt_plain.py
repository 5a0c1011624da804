"""Stale-data / race screen of the 128x160 kernel WITHOUT the LayerNorm ending (fp16, fp32, fp32 + residual) on grids of 1, 2 and 4 blocks per CU, short and long K: two operand sets
launched alternately, every fifth launch compared bit for bit with its set's first launch.  usage: python3 tools/soak_tt_plain.py [launches]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(0)
bad = 0
t0 = time.time()
for (M, N, K, kind) in [(8192, 1280, 1280, "f16"), (8192, 1280, 5120, "f32res"), (32768, 640, 640, "f16"), (32768, 640, 2560, "f32res"), (8192, 1280, 2560, "f32"), (4096, 1280, 5120, "f32res"),
                        (16384, 1280, 1280, "f32res"), (32768, 320, 960, "f32")]:
    sets = []
    for k in range(2):
        sets.append(dict(A=_lib.from_numpy((rng.standard_normal((M, K)) * (1 + k)).astype(np.float16)), R=_lib.from_numpy((rng.standard_normal((M, N)) + 3 * k).astype(np.float32)), C=_lib.DeviceBuffer(M * N * 4)))
    dW = _lib.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16))
    def mk(d):
        a = kernels.GemmArgs(A=d["A"].ptr, lda=K, W_=dW.ptr, ldb=K, M=M, N=N, K=K, tile_variant=31)
        if kind == "f16": a.C16, a.ldc16 = d["C"].ptr, N
        else: a.C32, a.ldc32 = d["C"].ptr, N
        if kind == "f32res": a.resid, a.ldr = d["R"].ptr, N
        return a
    args = [mk(d) for d in sets]
    name = kernels.gemm_variant(args[0])
    assert "128x160x64tt" in name, name
    nel = M * N // (2 if kind == "f16" else 1)
    first = []
    for k in range(2):
        kernels.gemm(args[k]); first.append(sets[k]["C"].download((nel,), np.uint32))
    bad0 = bad
    for r in range(reps):
        kernels.gemm(args[r & 1])
        if r % 10 >= 8 or r >= reps - 2:
            if not np.array_equal(sets[r & 1]["C"].download((nel,), np.uint32), first[r & 1]):
                bad += 1; print("MISMATCH", name, M, N, K, kind, "at launch", r, flush=True)
    print(f"{name} {M}x{N}x{K} {kind} ({(M // 128) * (N // 160)} blocks): {reps} launches on two alternating operand sets, every fifth checked: {'MISMATCHES' if bad > bad0 else 'ok'}", flush=True)
print("soak", "FAILED" if bad else "passed", f"in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
