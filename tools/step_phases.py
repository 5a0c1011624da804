"""Where one bench.py step spends its wall time: text conditioning, denoising loop, decode (host clock, stream synced between phases).
usage: python3 tools/step_phases.py [workload] [reps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import _lib, engine, text

wl = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model, w, h, B = {"sdxl": ("sdxl", 1024, 1024, 4), "sd15": ("sd1", 512, 512, 1), "tiny": ("tiny", 64, 64, 2)}[wl]
g = engine.Generator(model, w, h, B, n_step=20, cfg_scale=7.0, s_ancestral=1.0, use_hipgraph=(wl == "sd15"), weight_seed=1234)
tc = text.TextConditioner(model, w, h, seed=1234)
prompt = np.random.default_rng(7).integers(0, 49405 if model != "tiny" else 900, 8).astype(np.int32)
pp = prompt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
Lh = engine._proto2()
Lh.mlis_amd_textcond_apply.argtypes = [_lib.vp, _lib.vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
sync = lambda: engine.check1(Lh.mlis_amd_sync(g.h), "sync")
for r in range(reps + 1):
    t0 = time.perf_counter()
    engine.check1(Lh.mlis_amd_textcond_apply(tc.h, g.h, pp, prompt.size, None, 0), "textcond"); sync(); _lib.lib().mlsd_device_sync()
    t1 = time.perf_counter()
    g.denoise([42 + r * B + i for i in range(B)]); sync()
    t2 = time.perf_counter()
    g.decode()
    t3 = time.perf_counter()
    if r:
        print(f"{wl} b{B}: text cond {1e3 * (t1 - t0):7.2f} ms | denoise {1e3 * (t2 - t1):8.2f} ms (20 x 2B-batch UNet eval = {20 * g.last_unet_ms():8.2f} ms of it) | "
              f"decode {1e3 * (t3 - t2):7.2f} ms | total {1e3 * (t3 - t0):8.2f} ms")
