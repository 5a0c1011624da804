"""LayerNorm micro-benchmark (HIP events): one row per wave against the streaming form at several grid sizes."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, kernels
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ev = [vp(), vp()]
for e in ev: L.mlsd_event_create(ctypes.byref(e))
rng = np.random.default_rng(0)
for (M, d) in [(8192, 1280), (32768, 640), (8192, 320), (2048, 640)]:
    x = _lib.from_numpy(rng.standard_normal((M, d)).astype(np.float32))
    g, b = _lib.from_numpy(rng.standard_normal(d).astype(np.float32)), _lib.from_numpy(rng.standard_normal(d).astype(np.float32))
    y = _lib.DeviceBuffer(M * d * 2)
    ref = None
    for blocks in (0, 256, 512, 768, 1024, 2048):
        L.mlsd_layernorm_stream_blocks(blocks)
        fn = lambda: kernels.layernorm(x.ptr, d, M, d, 1e-5, g.ptr, b.ptr, y.ptr)
        for _ in range(5): fn()
        L.mlsd_event_record(ev[0], None)
        for _ in range(reps): fn()
        L.mlsd_event_record(ev[1], None); L.mlsd_event_sync(ev[1])
        ms = ctypes.c_float(); L.mlsd_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms))
        out = y.download((M, d), np.float16)
        if ref is None: ref = out
        t = ms.value / reps * 1e3
        print(f"layernorm {M}x{d} stream_blocks={blocks:5d}: {t:7.2f} us  {M*d*6/t/1e6:7.1f} TB/s  bit-equal to one-row-per-wave: {np.array_equal(out, ref)}")
L.mlsd_layernorm_stream_blocks(1024)
