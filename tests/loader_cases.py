"""Shared helpers of the loader tests: internal parameter lists of a model (from the engine's dry-mode plans) and a
synthetic single-file checkpoint written with CHECKPOINT-side names (tests/ckpt_names.py) by the `safetensors` package."""
import ctypes

import numpy as np

CLIP_OF = {"sd1": [("vit_l", "clip")], "sd2": [("vit_h", "clip")], "sdxl": [("vit_l", "clip"), ("vit_bigg", "clip2")],
           "tiny": [("tiny", "clip")], "tinyxl": [("tiny", "clip"), ("tiny", "clip2")]}


def model_params(model, lat=8):
    """[(internal name, is_f16, shape (torch order))] of UNet + VAE decoder + VAE encoder + text towers; dry-mode plans."""
    from mlimgsynth_amd import _lib, engine, text
    L = _lib.lib()
    was_dry = L.mlsd_runtime_is_dry()
    L.mlsd_runtime_dry(1)
    try:
        l = engine._proto2()
        out = []

        def add(ctx):
            for k, t, ne in ctx.param_list():
                out.append((k, t == 1, tuple(int(d) for d in ne[::-1])))
            ctx.destroy()                    # inside the dry block: its memory came from the dry runtime
        un = engine.Unet(model, lat, lat, 1, synth=False)
        add(un.ctx)
        vmodel = "sd1" if model == "sd2" else model
        P = engine.VaeParams()
        l.vae_params_get(vmodel.encode(), ctypes.byref(P))
        ctx, t = engine.MLCtx(), engine.vp()
        engine.check1(l.sdvae_decode_init(ctx.h, ctypes.byref(P), lat, lat, 1, ctypes.byref(t)), "init")
        engine.check1(l.sdvae_decode_build(ctx.h, ctypes.byref(P), t), "build")
        add(ctx)
        l.sdvae_encode_init.argtypes = [engine.vp, ctypes.POINTER(engine.VaeParams), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(engine.vp)]
        l.sdvae_encode_build.argtypes = [engine.vp, ctypes.POINTER(engine.VaeParams), engine.vp]
        ctx = engine.MLCtx()
        engine.check1(l.sdvae_encode_init(ctx.h, ctypes.byref(P), lat * 8, lat * 8, 1, ctypes.byref(t)), "init")
        engine.check1(l.sdvae_encode_build(ctx.h, ctypes.byref(P), t), "build")
        add(ctx)
        for tower, prefix in CLIP_OF[model]:
            enc = text.ClipEncoder.__new__(text.ClipEncoder)
            K = engine.ClipParams()
            l.clip_params_get(tower.encode(), ctypes.byref(K))
            ctx, E = engine.MLCtx(), text.ClipEncoderS()
            engine.check1(text._l().clip_encoder_init(ctypes.byref(E), ctx.h, ctypes.byref(K), prefix.encode(), 1, 1, True, True), "clip init")
            add(ctx)
        seen, uniq = set(), []
        for k, f, s in out:
            if k not in seen:
                seen.add(k)
                uniq.append((k, f, s))
        return uniq
    finally:
        L.mlsd_runtime_dry(1 if was_dry else 0)


def all_internal_names(model):
    return [k for k, _, _ in model_params(model)]


def synth_values(name, shape, f16, seed=1234):
    """the synthetic weight rule keyed by the INTERNAL name (so a loaded checkpoint equals mlctx_params_synth)"""
    import oracle_lib as O
    ne = (ctypes.c_int64 * 4)(*(list(shape)[::-1] + [1] * (4 - len(shape))))
    # squeeze trailing ones like the engine's shapes (they are stored 4-D)
    off, sc = ctypes.c_float(), ctypes.c_float()
    O.L().orc_synth_rule(name.encode(), 1 if f16 else 0, ctypes.byref(ne), ctypes.byref(off), ctypes.byref(sc))
    out = np.empty(int(np.prod(shape)), np.float32)
    O.L().orc_synth_fill(O.fptr(out), out.size, seed, name.encode(), off.value, sc.value, 1 if f16 else 0)
    return out.reshape(shape)


def squeeze_shape(shape):
    """engine shapes are 4-D with leading ones (torch order): strip them, keep conv kernels 4-D"""
    s = list(shape)
    while len(s) > 1 and s[0] == 1:
        s.pop(0)
    return tuple(s)


def write_checkpoint(path, model, dtype="F16", seed=1234, lat=8):
    """Single-file checkpoint with checkpoint-side names.  dtype: storage type of the F16-class tensors ('F16' | 'F32' | 'BF16');
    norm/bias/position tensors are stored F32 (as real SD checkpoints converted to fp16 still carry some)."""
    import ckpt_names as CN
    from safetensors.numpy import save_file
    tensors, fuse = {}, {}
    for k, f16, shape in model_params(model, lat):
        shp = squeeze_shape(shape) if not (len(shape) == 4 and shape[-1] <= 3 and shape[-2] <= 3 and k.endswith("weight") and shape[0] > 1) else shape
        v = synth_values(k, shape, f16, seed).reshape(shp)
        ext = CN.external_name(k, "sd1" if model == "tiny" else ("sdxl" if model == "tinyxl" else model))
        if f16 and dtype == "F16":
            v = v.astype(np.float16)
        if isinstance(ext, tuple):
            fuse.setdefault(ext[1], {})[ext[2]] = v
        else:
            if ext.endswith("text_projection") or ext.endswith("positional_embedding"):
                pass
            tensors[ext] = v
    for name, parts in fuse.items():
        tensors[name] = np.concatenate([parts[0], parts[1], parts[2]], axis=0)
    tensors["model_ema.decay"] = np.zeros(1, np.float32)          # a tensor the loader must ignore
    save_file(tensors, path, metadata={"format": "pt", "note": "synthetic test checkpoint"})
    return len(tensors)
