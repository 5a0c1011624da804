"""Small-Cout streaming convolution (conv_smalln.hip, tile variant 31) against the implicit-GEMM tile it replaces, on the VAE / TAESD output layers at full size.
usage: python3 tools/conv_smalln_bench.py [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
def timeit(fn):
    for _ in range(2): fn()
    L.mlsd_event_record(ev[0], None)
    for _ in range(reps): fn()
    L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
    ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
    return ms.value / reps
rng = np.random.default_rng(0)
for (n, h, w, cin, cout) in [(4, 1024, 1024, 128, 3), (4, 1024, 1024, 64, 3), (1, 512, 512, 128, 3), (1, 512, 512, 64, 3)]:
    M = n * h * w
    # several input sets rotated so that the activations come from HBM, not from the 256 MB MALL (4 x 1024 x 1024 x 128 halfs = 1.07 GB each anyway)
    nset = 2 if M * cin * 2 > (256 << 20) else 6
    xs = [_lib.from_numpy((rng.standard_normal((M, cin)) * 0.5).astype(np.float16)) for _ in range(nset)]
    wt = _lib.from_numpy((rng.standard_normal((cout, 9 * cin)) / np.sqrt(9 * cin)).astype(np.float16))
    bias = _lib.from_numpy(rng.standard_normal(cout).astype(np.float32))
    out = _lib.DeviceBuffer(M * cout * 4)
    it = [0]
    def launch(variant=0):
        x = xs[it[0] % nset]; it[0] += 1
        a = kernels.GemmArgs(A=x.ptr, lda=cin, conv=1, n_img=n, H=h, W=w, Cin=cin, OH=h, OW=w, KH=3, KW=3, stride=1, pad=1, upsample=0, W_=wt.ptr, ldb=9 * cin,
                             M=M, N=cout, K=9 * cin, bias=bias.ptr, rows_per_batch=h * w, ldrb=cout, C32=out.ptr, ldc32=cout, tile_variant=variant)
        kernels.gemm(a)
        return a
    gb = (M * cin * 2 + M * cout * 4) / 1e9
    ref = None
    for ring in (4, 5, 6):
        for strip in (8, 16, 32, 64, 128):
            L.mlsd_conv_smalln_set(ring, strip)
            t = timeit(launch)
            o = out.download((M, cout), np.float32)
            if ref is None: ref = o
            print(f"conv3x3 {n}x{h}x{w} cin {cin} cout {cout}  streaming ring {ring} strip {strip:3d}: {t*1e3:8.1f} us  {gb/t:6.2f} TB/s  max|diff| {np.abs(o-ref).max():.1e}", flush=True)
    L.mlsd_conv_smalln_set(0, 0)
    t = timeit(launch)
    print(f"conv3x3 {n}x{h}x{w} cin {cin} cout {cout}  streaming, automatic ring / strip      : {t*1e3:8.1f} us  {gb/t:6.2f} TB/s", flush=True)
    os.environ["MLSD_CONV_SMALLN"] = "0"
    # (the switch is read once per process: the GEMM tiles are timed by forcing the variant instead)
    for v, name in ((4, "256x128x32s3"), (0, "128x128x64s2")):
        L.mlsd_gemm_force_variant(v)
        t = timeit(launch)
        o = out.download((M, cout), np.float32)
        print(f"conv3x3 {n}x{h}x{w} cin {cin} cout {cout}  implicit GEMM {name:14s}      : {t*1e3:8.1f} us  {gb/t:6.2f} TB/s  max|diff| {np.abs(o-ref).max():.1e}", flush=True)
    L.mlsd_gemm_force_variant(-1)
