"""CPU only (oracle).  fp32-checkpoint deviation (BASELINE configs[0] is SD1.5 fp32; reference src/mlimgsynth.c:1235-1236 takes the linear weight type from the checkpoint): oracle with linear weights F32 (what the reference computes from an fp32 checkpoint) against linear weights F16 (what the device computes from the
same checkpoint: weights rounded on load)."""
import sys, ctypes, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import oracle_lib as O
L = O.L()
L.orc_set_linear_wtype.argtypes = [ctypes.c_int]
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for model, lat, steps in [("sd1", 32, 20), ("sdxl", 32, 20)]:
    U = O.unet_params(model)
    rng = np.random.default_rng(3)
    cond = rng.standard_normal((77, U.n_ctx)).astype(np.float32); uncond = (cond * 0.25).astype(np.float32)
    label = rng.standard_normal(U.ch_adm_in).astype(np.float32) if U.ch_adm_in else None
    outs = {}
    for wt in (1, 0):
        L.orc_set_linear_wtype(wt)
        P = O.Params(1234)
        out = np.empty((4, lat, lat), np.float32); tu = ctypes.c_double()
        # one evaluation first
        x = rng.standard_normal((1, 4, lat, lat)).astype(np.float32) if wt == 1 else x
        e1 = O.from_ot(L.orc_unet_denoise_run(P.h, b"unet", U, O.to_ot(x), O.to_ot(cond[None, None]), O.to_ot(label[None, None, None]) if label is not None else None, 5.0))
        nfe = L.orc_generate_latent(P.h, b"unet", U, lat, lat, O.to_ot(cond[None, None]), O.to_ot(label[None, None, None]) if label is not None else None,
                                    O.to_ot(uncond[None, None]), O.to_ot(label[None, None, None]) if label is not None else None, 7.0, steps, 1.0, 42, 0, O.fptr(out), ctypes.byref(tu))
        outs[wt] = (e1.copy(), out.copy()); P.free()
    print(f"{model} latent {lat}: one evaluation F16-linear vs F32-linear rel-L2 {rel(outs[1][0], outs[0][0]):.2e}; {steps}-step final latent {rel(outs[1][1], outs[0][1]):.2e}")
L.orc_set_linear_wtype(1)
