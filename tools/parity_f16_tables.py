"""HIP path against the oracle in BOTH activation modes (exact fp32 GELU / quick-GELU, and ggml-CPU's F16 lookup tables: orc_set_ggml_f16_tables): one UNet evaluation
of the tiny / tinyxl / real SD1.5 (16x16 latent) / real SDXL (16x16) tables and the CLIP towers.  usage (GPU box): python3 tools/parity_f16_tables.py > profiles/r4_parity_f16_tables.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from mlimgsynth_amd import engine


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def both(fn):
    r0 = fn()
    O.L().orc_set_ggml_f16_tables(1)
    try: r1 = fn()
    finally: O.L().orc_set_ggml_f16_tables(0)
    return r0, r1


print("# HIP path vs oracle (rel-L2), oracle mode: exact fp32 activations | ggml-CPU F16 lookup tables (GELU, quick-GELU); last column: the two oracle modes against each other")
print(f"# {'case':44s} {'vs exact':>10s} {'vs f16-table':>12s} {'modes differ':>12s}")
for model, lat, n in (("tiny", 8, 2), ("tinyxl", 8, 2), ("sd1", 16, 1), ("sdxl", 16, 1)):
    rng = np.random.default_rng(5)
    un = engine.Unet(model, lat, lat, n)
    P = un.P
    x = rng.standard_normal((n, 4, lat, lat)).astype(np.float32) * 4
    cond = rng.standard_normal((n, 77, P.n_ctx)).astype(np.float32)
    label = rng.standard_normal((n, P.ch_adm_in)).astype(np.float32) if P.ch_adm_in else None
    sigma = np.array([7.0, 0.5][:n], np.float32)
    got = un.run(x, cond, label, sigma)
    U = O.unet_params(model)

    def ora():
        OP = O.Params(1234)
        return np.stack([O.from_ot(O.L().orc_unet_denoise_run(OP.h, b"unet", U, O.to_ot(x[i:i + 1]), O.to_ot(cond[i][None, None]),
                                                              O.to_ot(label[i][None, None, None]) if label is not None else None, float(sigma[i])))[0] for i in range(n)])
    r0, r1 = both(ora)
    print(f"{'unet ' + model + f' latent {lat} (GEGLU gate: GELU)':46s} {rel(got, r0):10.3e} {rel(got, r1):12.3e} {rel(r1, r0):12.3e}", flush=True)
    un.ctx.destroy()
for model, prefix, skip, norm in (("tiny", "clip", 1, True), ("vit_l", "clip", 1, True), ("vit_bigg", "clip2", 2, False)):
    try:
        K = O.clip_params(model)
    except Exception as e:
        print(f"clip {model}: skipped ({e})"); continue
    rng = np.random.default_rng(6)
    n_tok = 9
    toks = rng.integers(0, K.n_vocab - 3, (1, n_tok)).astype(np.int32)
    emb, _ = engine.clip_text_encode(model, prefix, toks, want_embed=True, want_feat=False, clip_skip=skip, norm=norm)
    full = np.full(K.n_token, K.tok_pad, np.int32); full[0] = K.tok_start; full[1:1 + n_tok] = toks[0]; full[1 + n_tok] = K.tok_end
    ptr = full.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))

    def ora():
        OP = O.Params(1234)
        return O.from_ot(O.L().orc_clip_text_encode(OP.h, prefix.encode(), K, ptr, skip, int(norm), 0, 0)).reshape(K.n_token, K.d_embed)
    r0, r1 = both(ora)
    act = "GELU" if K.d_embed in (1024, 1280) else "quick-GELU"
    print(f"{'clip ' + model + ' (' + act + ')':46s} {rel(emb[0], r0):10.3e} {rel(emb[0], r1):12.3e} {rel(r1, r0):12.3e}", flush=True)
