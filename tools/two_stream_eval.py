"""Experiment: one UNet evaluation of batch N on one stream against two evaluations of batch N/2 on two streams (the cond and
uncond halves of a cfg batch are independent).  Wall time per N inputs, HIP events on a third stream are not needed: host clock
around enqueue + sync of 10 repetitions.
usage: python3 tools/two_stream_eval.py [model] [latent] [N] [reps] [plan flags]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, engine
L = _lib.lib(); vp = _lib.vp
model = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # 8 = MLB_F_HIPGRAPH (each plan replayed as one graph launch)

def stream():
    s = vp(); _lib.check(L.mlsd_stream_create(ctypes.byref(s)), "stream"); return s.value

s0, s1, s2 = stream(), stream(), stream()
full = engine.Unet(model, lat, lat, N, stream=s0, flags=flags)
ha, hb = engine.Unet(model, lat, lat, N // 2, stream=s1, flags=flags), engine.Unet(model, lat, lat, N // 2, stream=s2, flags=flags)
print("tile-table misses:", L.mlctx_tune_misses())

def timed(fn, sync):
    for _ in range(2): fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3

t_full = timed(lambda: full.ctx.compute(), lambda: full.ctx.sync())
t_half = timed(lambda: ha.ctx.compute(), lambda: ha.ctx.sync())
def both():
    ha.ctx.compute(); hb.ctx.compute()
def sync_both():
    ha.ctx.sync(); hb.ctx.sync()
t_two = timed(both, sync_both)
print(f"{model} latent {lat}: one stream, batch {N}: {t_full:.2f} ms | one stream, batch {N//2}: {t_half:.2f} ms (x2 = {2*t_half:.2f}) | "
      f"two streams, batch {N//2} each: {t_two:.2f} ms")
