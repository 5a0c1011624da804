"""Generates tests/golden/torch_golden.npz: expected outputs of the golden cases (tests/golden_cases.py) computed by the
INDEPENDENT torch restatement (tools/torch_ref.py, fp32 CPU, ggml's F16 operand rounding emulated).  Run in the build
container only (torch CPU); prints, for information, how far the oracle is from each vector.

usage: python3 tools/make_torch_golden.py [--only key_prefix] [--check]     (--check: compare, do not write)
       python3 tools/make_torch_golden.py --headline [--only key_prefix]   (full-size cases -> torch_golden_headline.npz)
Shared DATA only: parameter names + the (seed, name, shape) synthetic weight generator, the pinned host scalars
(sigma<->t table and Philox noise, both pinned against the reference itself: tests/test_oracle_host.py)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as G      # noqa: E402
import oracle_lib as O        # noqa: E402   (weight generator + pinned host scalars + the informational comparison)
from tools import torch_ref as TR   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "torch_golden.npz")
torch.set_num_threads(os.cpu_count() or 8)
torch.manual_seed(0)


def synth(name, shape, f16):
    """the synthetic weight rule (oracle/o_core.c orc_synth_rule / orc_synth_fill == mlctx_params_synth): DATA"""
    ne = (ctypes.c_int64 * 4)(*(list(shape)[::-1] + [1] * (4 - len(shape))))
    off, sc = ctypes.c_float(), ctypes.c_float()
    O.L().orc_synth_rule(name.encode(), 1 if f16 else 0, ctypes.byref(ne), ctypes.byref(off), ctypes.byref(sc))
    out = np.empty(int(np.prod(shape)), np.float32)
    O.L().orc_synth_fill(O.fptr(out), out.size, G.WEIGHT_SEED, name.encode(), off.value, sc.value, 1 if f16 else 0)
    return out.reshape(shape)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def run_mode(out, want, check, f16_mode):
    SUF = "" if f16_mode else "__f32"      # "__f32": no F16 operand rounding on either side (graph check at 1e-5 class)
    O.L().orc_set_act_rounding(1 if f16_mode else 0)
    NetM = lambda: TR.Net(TR.Weights(synth), f16_ops=f16_mode)
    print("==== mode:", "ggml F16 operand rounding" if f16_mode else "pure fp32 operands", flush=True)
    with torch.no_grad():
        for key, model, lat, n, sigmas in G.UNET_CASES:
            if not want(key):
                continue
            t0 = time.time()
            U = G.UNET[model]
            x, cond, label = G.unet_inputs(key, model, lat, n)
            net = NetM()
            res = []
            for i in range(n):
                s = np.float32(sigmas[i])
                t = O.L().orc_sigma_to_t(float(s))                   # pinned host scalar (reference-produced table)
                c_in = np.float32(1) / np.sqrt(s * s + np.float32(1), dtype=np.float32)
                y = net.unet(U, torch.from_numpy(x[i:i + 1] * c_in), torch.tensor([t]), torch.from_numpy(cond[i:i + 1]),
                             torch.from_numpy(label[i:i + 1]) if label is not None else None)
                res.append(y[0].numpy())
            res = np.stack(res)
            # informational: the oracle on the same inputs
            UO, P = O.unet_params(model), O.Params(G.WEIGHT_SEED)
            errs = []
            for i in range(n):
                lab = O.to_ot(label[i][None, None, None]) if label is not None else None
                r = O.from_ot(O.L().orc_unet_denoise_run(P.h, b"unet", UO, O.to_ot(x[i:i + 1]), O.to_ot(cond[i][None, None]), lab, float(sigmas[i])))[0]
                errs.append(rel(r, res[i]))
            P.free()
            print(f"{key}: torch {time.time() - t0:.1f}s, oracle-vs-torch rel-L2 {errs}", flush=True)
            if check:
                print("   stored-vs-new", rel(out[key + SUF], res))
            out[key + SUF] = res

        for key, model, lat in G.VAE_CASES:
            if not want(key):
                continue
            z = G.vae_inputs(key, lat)
            net = NetM()
            res = net.vae_decode(G.VAE[model], torch.from_numpy(z)).numpy()
            P = O.Params(G.WEIGHT_SEED)
            r = O.from_ot(O.L().orc_vae_decode(P.h, b"vae", O.vae_params(model), O.to_ot(z)))
            P.free()
            print(f"{key}: oracle-vs-torch rel-L2 {rel(r - 0.5, res - 0.5)}", flush=True)
            out[key + SUF] = res

        for key, lat in G.TAE_CASES:
            if not want(key):
                continue
            z = G.tae_inputs(key, lat)
            res = NetM().tae_decode(torch.from_numpy(z)).numpy()
            P = O.Params(G.WEIGHT_SEED)
            r = O.from_ot(O.L().orc_tae_decode(P.h, b"tae", O.to_ot(z)))
            P.free()
            print(f"{key}: oracle-vs-torch rel-L2 {rel(r, res)}", flush=True)
            out[key + SUF] = res

        for key, model, prefix, skip, norm, feat, n_tok in G.CLIP_CASES:
            if not want(key):
                continue
            K = G.CLIP[model]
            toks, full = G.clip_tokens(key, model, n_tok)
            net = NetM()
            emb = net.clip_text(K, torch.from_numpy(full[None]), prefix, skip, norm)[0].numpy()
            out[key + SUF] = emb
            P = O.Params(G.WEIGHT_SEED)
            KO = O.clip_params(model)
            ptr = full.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            r = O.from_ot(O.L().orc_clip_text_encode(P.h, prefix.encode(), KO, ptr, skip, int(norm), 0, 0)).reshape(K["n_token"], K["d_embed"])
            msg = f"{key}: oracle-vs-torch embed rel-L2 {rel(r, emb)}"
            if feat:
                ft = net.clip_feat(K, torch.from_numpy(full[None]), prefix, n_tok + 1)[0].numpy()
                out[key + "_feat" + SUF] = ft
                rf = O.from_ot(O.L().orc_clip_text_encode(P.h, prefix.encode(), KO, ptr, -1, 1, 1, n_tok + 1)).reshape(K["d_embed"])
                msg += f", feat {rel(rf, ft)}"
                out[key + "_label" + SUF] = TR.sdxl_label(torch.from_numpy(ft), 1024, 768).numpy()
            P.free()
            print(msg, flush=True)

        for key, model, lat, steps, seed in G.GEN_CASES:
            if not want(key):
                continue
            U = G.UNET[model]
            cond, uncond, label, unlabel = G.gen_inputs(key, model)
            net = NetM()
            sig = np.zeros(steps + 2, np.float32)
            O.L().orc_schedule(steps, 1, 1.0, 0.0, O.fptr(sig))     # pinned schedule (SURVEY row a13 golden)
            sig = sig[:steps + 1]
            per = 4 * lat * lat

            def eps_cfg(x, sigma, cfg=7.0):
                t = O.L().orc_sigma_to_t(float(sigma))
                c_in = 1.0 / np.sqrt(np.float32(sigma) ** 2 + 1, dtype=np.float32)
                xi = (x * c_in).float()
                ec = net.unet(U, xi, torch.tensor([t]), torch.from_numpy(cond[None]), torch.from_numpy(label[None]) if label is not None else None)
                eu = net.unet(U, xi, torch.tensor([t]), torch.from_numpy(uncond[None]), torch.from_numpy(unlabel[None]) if unlabel is not None else None)
                return ec * cfg + eu * (1 - cfg)
            noise = lambda i: torch.from_numpy(O.randn(seed, i, per).reshape(1, 4, lat, lat))   # Philox: pinned vs the reference build
            res = TR.euler_ancestral(eps_cfg, noise(0), sig, noise)[0].numpy()
            P = O.Params(G.WEIGHT_SEED)
            ref = np.empty((4, lat, lat), np.float32)
            tu = ctypes.c_double()
            O.L().orc_generate_latent(P.h, b"unet", O.unet_params(model), lat, lat, O.to_ot(cond[None, None]),
                                      O.to_ot(label[None, None, None]) if label is not None else None, O.to_ot(uncond[None, None]),
                                      O.to_ot(unlabel[None, None, None]) if unlabel is not None else None,
                                      7.0, steps, 1.0, seed, 0, O.fptr(ref), ctypes.byref(tu))
            P.free()
            print(f"{key}: oracle-vs-torch final latent rel-L2 {rel(ref, res)}", flush=True)
            out[key + SUF] = res

        for key, model, side in G.VAE_ENC_CASES:
            if not want(key):
                continue
            img = G.image_inputs(key, side)
            res = NetM().vae_encode_moments(G.VAE[model], torch.from_numpy(img)).numpy()
            out[key + SUF] = res
            if hasattr(O.L(), "orc_vae_encode_moments"):
                P = O.Params(G.WEIGHT_SEED)
                r = O.from_ot(O.L().orc_vae_encode_moments(P.h, b"vae", O.vae_params(model), O.to_ot(img)))
                P.free()
                print(f"{key}: oracle-vs-torch moments rel-L2 {rel(r, res)}", flush=True)
            else:
                print(f"{key}: generated (no oracle encoder yet)")

    O.L().orc_set_act_rounding(1)


def run_headline(want):
    """full-size cases -> tests/golden/torch_golden_headline.npz (fp16-operand mode; images reduced by G.reduce_image)"""
    path = os.path.join(ROOT, "tests", "golden", "torch_golden_headline.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    O.L().orc_set_act_rounding(1)
    NetM = lambda: TR.Net(TR.Weights(synth), f16_ops=True)
    with torch.no_grad():
        for key, model, lat, n, sigmas in G.HEADLINE_UNET_CASES:
            if not want(key):
                continue
            t0 = time.time()
            U = G.UNET[model]
            x, cond, label = G.unet_inputs(key, model, lat, n)
            s = np.float32(sigmas[0])
            t = O.L().orc_sigma_to_t(float(s))
            c_in = np.float32(1) / np.sqrt(s * s + np.float32(1), dtype=np.float32)
            y = NetM().unet(U, torch.from_numpy(x[:1] * c_in), torch.tensor([t]), torch.from_numpy(cond[:1]),
                            torch.from_numpy(label[:1]) if label is not None else None)
            out[key] = y.numpy()
            print(f"{key}: torch {time.time() - t0:.1f}s", flush=True)
        for key, model, lat in G.HEADLINE_VAE_CASES:
            if not want(key):
                continue
            t0 = time.time()
            res = NetM().vae_decode(G.VAE[model], torch.from_numpy(G.vae_inputs(key, lat))).numpy()
            out[key] = G.reduce_image(res, key)
            print(f"{key}: torch {time.time() - t0:.1f}s, image {res.shape} -> {out[key].shape}", flush=True)
        for key, lat in G.HEADLINE_TAE_CASES:
            if not want(key):
                continue
            t0 = time.time()
            res = NetM().tae_decode(torch.from_numpy(G.tae_inputs(key, lat))).numpy()
            out[key] = G.reduce_image(res, key)
            print(f"{key}: torch {time.time() - t0:.1f}s", flush=True)
    np.savez_compressed(path, **{k: np.asarray(v, np.float32) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "vectors")


def main():
    only = None
    for i, a in enumerate(sys.argv):
        if a == "--only":
            only = sys.argv[i + 1]
    check = "--check" in sys.argv
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    want = lambda k: only is None or k.startswith(only)
    if "--headline" in sys.argv:
        return run_headline(want)

    for f16_mode in (True, False):
        run_mode(out, want, check, f16_mode)

    if not check:
        np.savez_compressed(OUT, **{k: np.asarray(v, np.float32) for k, v in out.items()})
        print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "vectors")


if __name__ == "__main__":
    main()
