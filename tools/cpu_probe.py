"""Host-CPU probe for the bench's CPU baseline (oracle SGEMM rate against the OpenMP thread count; the SD1.5 sample's parts).  usage: python3 tools/cpu_probe.py
Measured on the GPU box (2 x EPYC 9575F, 256 logical CPUs, cgroup quota 16 CPUs; profiles/r5_cpu_probe.txt): 16 threads 2.9 TFLOP/s, 32: 5.1, 64: 7.4 in short bursts."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
L = O.L()
print("host cpus", os.cpu_count(), "isa", L.orc_get_isa(), flush=True)
os.system("lscpu | grep -E 'Model name|Socket|NUMA node\\(s\\)|Thread' | head -5")
def run(M, N, K, reps=2):
    A = np.random.randn(M, K).astype(np.float32); B = np.random.randn(N, K).astype(np.float32); C = np.zeros((M, N), np.float32)
    L.orc_sgemm_nt(M, N, K, O.fptr(A), K, O.fptr(B), K, O.fptr(C), N)
    best = 1e9
    for _ in range(reps):
        t = time.time(); L.orc_sgemm_nt(M, N, K, O.fptr(A), K, O.fptr(B), K, O.fptr(C), N); best = min(best, time.time() - t)
    return 2.0 * M * N * K / best / 1e9
for th in (8, 16, 32, 64):      # (never more: on the GPU box -- cgroup quota 16 CPUs of 256 -- 128 threads ran at 1.3 TFLOP/s and 256 at 0.06: spinning in barriers; the 256-thread pass alone took 15 minutes)
    if th > (os.cpu_count() or 1): break
    L.orc_set_threads(th)
    print(f"threads {th:3d}: 4096x4096x3840 {run(4096, 4096, 3840):8.1f} GFLOP/s   8192x1280x5120 {run(8192, 1280, 5120):8.1f}   1280x16384x2880 {run(1280, 16384, 2880):8.1f}   320x4096x2880 {run(320, 4096, 2880):8.1f}", flush=True)
import bench
from mlimgsynth_amd import _lib, engine
_lib.lib().mlsd_runtime_dry(1)
un = engine.Unet("sd1", 64, 64, 2, synth=False)
plist = un.ctx.param_list()
for th in (bench.host_threads(), 32):
    t0 = time.time()
    rec, OP, V, Oo = bench.cpu_sample_e2e("sd1", 512, 512, 7.0, 40, th, plist)
    t1 = time.time()
    bench.cpu_decode_sample(rec, OP, V, Oo, 512, 512, 0.803e12, 2.515e12, 120.0)
    print(f"threads {th}: SD1.5 sample {rec}; wall {t1 - t0:.1f} s (incl. weight synthesis) + decode {time.time() - t1:.1f} s", flush=True)
    L.orc_prof_dump(1)
    OP.free()
