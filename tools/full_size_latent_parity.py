"""BASELINE.json configs[0]/[1] end to end: SD1.5 512x512, 20-step Euler-a, cfg 7, seed 42 -- the HIP engine against the CPU
oracle (fp32 restatement of the reference path, OpenMP) on the same synthetic weights (seed 1234) and conditioning.  Reports the
rel-L2 of the FINAL latent after 40 UNet evaluations ("reported, not assumed").  Minutes of CPU time on
the GPU box's host cores: a tool, not a test.   usage: python3 tools/full_size_latent_parity.py [model] [latent] [steps]"""
import ctypes, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as O
from mlimgsynth_amd import engine

model = sys.argv[1] if len(sys.argv) > 1 else "sd1"
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
U = O.unet_params(model)
rng = np.random.default_rng(8)
cond = rng.standard_normal((77, U.n_ctx)).astype(np.float32)
uncond = np.zeros_like(cond) if U.uncond_empty_zero else rng.standard_normal((77, U.n_ctx)).astype(np.float32)
label = rng.standard_normal(U.ch_adm_in).astype(np.float32) if U.ch_adm_in else None
unlabel = rng.standard_normal(U.ch_adm_in).astype(np.float32) if U.ch_adm_in else None
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
t0 = time.time()
g = engine.Generator(model, lat * 8, lat * 8, 1, n_step=steps, cfg_scale=7.0, s_ancestral=1.0)
g.set_cond(cond, label, uncond, unlabel)
got, _ = g.generate([42], want_images=False)
t_gpu = time.time() - t0
t0 = time.time()
threads = min(os.cpu_count() or 1, 64)          # the oracle's OpenMP loops stop scaling (and then slow down) past the physical cores of a socket
O.L().orc_set_threads(threads)
P = O.Params(1234)
out = np.empty((4, lat, lat), np.float32)
tu = ctypes.c_double()
nfe = O.L().orc_generate_latent(P.h, b"unet", U, lat, lat, O.to_ot(cond[None, None]),
                                O.to_ot(label[None, None, None]) if label is not None else None,
                                O.to_ot(uncond[None, None]), O.to_ot(unlabel[None, None, None]) if unlabel is not None else None,
                                7.0, steps, 1.0, 42, 0, O.fptr(out), ctypes.byref(tu))
t_cpu = time.time() - t0
print(f"{model} latent {lat}x{lat}, {steps}-step Euler-a, cfg 7, seed 42, {nfe} UNet evaluations: final latent rel-L2 (HIP fp16 MFMA vs fp32 CPU oracle) = "
      f"{rel(got[0], out.astype(np.float64)):.3e}; max |latent| {np.abs(out).max():.2f}; finite {bool(np.isfinite(got).all())}; "
      f"engine incl. setup {t_gpu:.1f} s, oracle {t_cpu:.1f} s ({threads} OpenMP threads)")
