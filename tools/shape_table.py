"""Per-SHAPE kernel time table of one UNet / VAE-decoder evaluation (per-launch HIP events, mlctx_profile_ops).
usage: python3 tools/shape_table.py <model> <latent_side> <n_batch> [unet|vae|tae] [reps]
GEMM labels carry MxNxK (MLB_F_OPSHAPES); rows are sorted by total time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlimgsynth_amd import engine

MLB_F_OPSHAPES = 16
model, lat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
what = sys.argv[4] if len(sys.argv) > 4 else "unet"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
if what == "unet":
    obj = engine.Unet(model, lat, lat, n, flags=MLB_F_OPSHAPES)
    ctx = obj.ctx
else:
    obj = engine.Decoder(model, lat, lat, n, tae=(what == "tae"), flags=MLB_F_OPSHAPES)
    ctx = obj.ctx
ctx.compute(); ctx.sync()
ops = ctx.op_list()
nbytes = ctx.op_bytes()
ms = np.min([ctx.profile_ops() for _ in range(reps)], axis=0)
agg = {}
for (lab, fl), t, nb in zip(ops, ms, nbytes):
    if t == 0.0: continue                     # a norm that runs inside its producer (mlctx_profile_ops reports 0 for it): not a launch
    e = agg.setdefault(lab, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += float(t); e[2] += fl; e[3] += nb
print(f"# {what} {model} latent {lat} batch {n}: {int((ms > 0).sum())} launches, {ms.sum():.3f} ms (sum of per-launch HIP-event times, min of {reps})")
print(f"# {'kernel':60s} {'n':>4s} {'total_ms':>9s} {'us/launch':>9s} {'TFLOP/s':>8s} {'GB/s':>8s}")
for lab, (c, t, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{lab:62s} {c:4d} {t:9.3f} {t / c * 1e3:9.1f} {fl / (t * 1e-3) / 1e12 if t > 0 else 0:8.1f} {nb / (t * 1e-3) / 1e9 if t > 0 else 0:8.1f}")
