"""Experiment: does the VAE decode of one batch run in the shadow of the next batch's denoising loop?  The decode is mostly bound by its fp32 activations (HBM), the UNet
evaluation by the matrix pipes.  20 evaluations of the SDXL batch-8 plan on one stream and one KL-VAE decode of 4 images on another: one after the other against side by side.
usage: python3 tools/unet_vae_overlap.py [reps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlimgsynth_amd import _lib, engine
L = _lib.lib(); vp = _lib.vp
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def stream():
    s = vp(); _lib.check(L.mlsd_stream_create(ctypes.byref(s)), "stream"); return s.value
s1, s2 = stream(), stream()
un = engine.Unet("sdxl", 128, 128, 8, stream=s1)
de = engine.Decoder("sdxl", 128, 128, 4, stream=s2)
for _ in range(2): un.ctx.compute(); de.ctx.compute()
un.ctx.sync(); de.ctx.sync()
def timed(fn):
    best = 1e9
    for _ in range(reps):
        un.ctx.sync(); de.ctx.sync()
        t0 = time.perf_counter(); fn(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
def denoise():
    for _ in range(20): un.ctx.compute()
t_un = timed(lambda: (denoise(), un.ctx.sync()))
t_de = timed(lambda: (de.ctx.compute(), de.ctx.sync()))
def seq():
    denoise(); un.ctx.sync(); de.ctx.compute(); de.ctx.sync()
def par():
    de.ctx.compute(); denoise(); un.ctx.sync(); de.ctx.sync()
t_seq, t_par = timed(seq), timed(par)
print(f"20 evaluations {t_un:.1f} ms | decode {t_de:.1f} ms | one after the other {t_seq:.1f} ms | decode on a second stream beside the evaluations {t_par:.1f} ms "
      f"({t_seq - t_par:+.1f} ms = {100 * (t_seq - t_par) / t_seq:.1f} % of a step)")
