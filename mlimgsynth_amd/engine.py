"""Python (ctypes) mirror of the host-side C API (include/mlblock_amd.h, include/mlimgsynth_amd.h).

Thin by design: every call goes straight into libmlimgsynth_amd.so; numpy arrays at the
boundary use the reference's host layout (LocalTensor: NCHW fp32, src/localtensor.h:16-20).
"""
import ctypes

import numpy as np

from ._lib import check1, lib, vp

c_int, c_f, c_i64, c_u64 = ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_uint64
FP = ctypes.POINTER(ctypes.c_float)


class UnetParams(ctypes.Structure):
    _fields_ = [("n_ch_in", c_int), ("n_ch_out", c_int), ("n_res_blk", c_int), ("attn_res", c_int * 4),
                ("ch_mult", c_int * 5), ("transf_depth", c_int * 5), ("n_te", c_int), ("n_head", c_int),
                ("d_head", c_int), ("n_ctx", c_int), ("n_ch", c_int), ("ch_adm_in", c_int),
                ("clip_norm", c_int), ("cond_label", c_int), ("uncond_empty_zero", c_int), ("vparam", c_int),
                ("n_step_train", c_int), ("sigma_min", c_f), ("sigma_max", c_f)]


class UnetState(ctypes.Structure):
    _fields_ = [("ctx", vp), ("par", ctypes.POINTER(UnetParams)), ("nfe", ctypes.c_uint), ("lw", c_int), ("lh", c_int),
                ("n_batch", c_int), ("t_x", vp), ("t_t", vp), ("t_c", vp), ("t_l", vp), ("t_out", vp)]


class VaeParams(ctypes.Structure):
    _fields_ = [("ch_x", c_int), ("ch_z", c_int), ("ch", c_int), ("n_res", c_int), ("n_res_blk", c_int),
                ("ch_mult", c_int * 5), ("d_embed", c_int), ("f_down", c_int), ("scale_factor", c_f)]


class ClipParams(ctypes.Structure):
    _fields_ = [("n_vocab", c_int), ("n_token", c_int), ("d_embed", c_int), ("n_interm", c_int),
                ("n_head", c_int), ("n_layer", c_int), ("tok_start", c_int), ("tok_end", c_int), ("tok_pad", c_int)]


class CtxInfo(ctypes.Structure):
    _fields_ = [("mem_params", ctypes.c_size_t), ("mem_compute", ctypes.c_size_t), ("mem_total", ctypes.c_size_t),
                ("t_load", ctypes.c_double), ("t_compute", ctypes.c_double), ("n_compute", ctypes.c_uint),
                ("n_conv", ctypes.c_uint), ("n_ops", ctypes.c_uint), ("flops", ctypes.c_double)]


class AmdConfig(ctypes.Structure):
    _fields_ = [("model", ctypes.c_char_p), ("width", c_int), ("height", c_int), ("n_batch", c_int), ("n_step", c_int),
                ("cfg_scale", c_f), ("s_ancestral", c_f), ("sched", c_int), ("use_tae", c_int), ("use_hipgraph", c_int),
                ("weight_seed", c_u64), ("method", c_int), ("s_noise", c_f), ("f_t_ini", c_f), ("f_t_end", c_f),
                ("defer_weights", c_int), ("unet_split", c_int)]


_proto_done = False


def L():
    global _proto_done
    l = lib()
    if not _proto_done:
        l.mlctx_new.restype = vp
        l.mlctx_new.argtypes = [vp]
        l.mlctx_destroy.argtypes = [vp]
        l.mlctx_set_flags.argtypes = [vp, c_int]
        l.mlctx_params_synth.argtypes = [vp, c_u64]
        l.mlctx_param_set.argtypes = [vp, ctypes.c_char_p, c_int, vp, c_i64]
        l.mlctx_param_count.argtypes = [vp]
        l.mlctx_param_info.argtypes = [vp, c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_int), ctypes.POINTER(c_i64 * 4)]
        l.mlctx_info.argtypes = [vp, ctypes.POINTER(CtxInfo)]
        l.mlctx_op_info.argtypes = [vp, c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double)]
        l.mlctx_profile_ops.argtypes = [vp, FP, c_int]
        l.mlctx_op_bytes.argtypes = [vp, c_int]
        l.mlctx_op_bytes.restype = ctypes.c_double
        l.mlctx_compute.argtypes = [vp]
        l.mlctx_sync.argtypes = [vp]
        l.unet_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(UnetParams)]
        l.unet_denoise_init_n.argtypes = [ctypes.POINTER(UnetState), vp, ctypes.POINTER(UnetParams), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
        l.unet_denoise_build.argtypes = [ctypes.POINTER(UnetState)]
        l.unet_denoise_run_n.argtypes = [ctypes.POINTER(UnetState), FP, FP, FP, FP, FP]
        l.unet_sigma_to_t.restype = c_f
        l.unet_sigma_to_t.argtypes = [ctypes.POINTER(UnetParams), c_f]
        l.unet_t_to_sigma.restype = c_f
        l.unet_t_to_sigma.argtypes = [ctypes.POINTER(UnetParams), c_f]
        l.mlb_f32_to_f16_bits.restype = ctypes.c_uint16
        l.mlb_f32_to_f16_bits.argtypes = [c_f]
        l.mlb_f16_bits_to_f32.restype = c_f
        l.mlb_f16_bits_to_f32.argtypes = [ctypes.c_uint16]
        _proto_done = True
    return l


def fptr(a):
    return a.ctypes.data_as(FP) if a is not None else None


class MLCtx:
    """Owns one MLCtx (one model graph + its device-resident parameters)."""

    def __init__(self, stream=None, flags=0):
        self.h = L().mlctx_new(vp(stream))
        if flags:
            L().mlctx_set_flags(self.h, flags)

    def params_synth(self, seed=1234):
        check1(L().mlctx_params_synth(self.h, seed), "mlctx_params_synth")

    def param_set(self, key, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        check1(L().mlctx_param_set(self.h, key.encode(), 0, a.ctypes.data_as(vp), a.size), f"mlctx_param_set({key})")

    def param_list(self):
        out = []
        for i in range(L().mlctx_param_count(self.h)):
            key, typ, ne = ctypes.c_char_p(), c_int(), (c_i64 * 4)()
            L().mlctx_param_info(self.h, i, ctypes.byref(key), ctypes.byref(typ), ctypes.byref(ne))
            out.append((key.value.decode(), typ.value, [int(v) for v in ne]))
        return out

    def info(self):
        i = CtxInfo()
        L().mlctx_info(self.h, ctypes.byref(i))
        return i

    def op_list(self):
        out = []
        n = self.info().n_ops
        for i in range(n):
            lab, fl = ctypes.c_char_p(), ctypes.c_double()
            L().mlctx_op_info(self.h, i, ctypes.byref(lab), ctypes.byref(fl))
            out.append((lab.value.decode(), fl.value))
        return out

    def op_bytes(self):
        f = L().mlctx_op_bytes
        f.restype = ctypes.c_double
        return [f(self.h, i) for i in range(self.info().n_ops)]

    def profile_ops(self):
        n = self.info().n_ops
        ms = np.zeros(n, np.float32)
        check1(L().mlctx_profile_ops(self.h, fptr(ms), n), "mlctx_profile_ops")
        return ms

    def streaming_info(self):
        """(segments, streamed bytes per evaluation, slab bytes, host bytes) of a weight-streaming plan, or None"""
        f = L().mlctx_weight_streaming_info
        f.argtypes = [vp, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        n, a, b, c = c_int(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        if not f(self.h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)):
            return None
        return n.value, a.value, b.value, c.value

    def streaming_copies(self):
        """host -> device copies one evaluation of a weight-streaming plan issues (the host master is laid out in segment order: one per segment at best)"""
        f = L().mlctx_weight_streaming_copies
        f.argtypes = [vp]
        return int(f(self.h))

    def tune_misses(self):
        """GEMM shapes of this plan that the compiled-in tile table does not list (they run on the static rule)."""
        f = L().mlctx_plan_tune_misses
        f.argtypes = [vp]
        return int(f(self.h))

    def tune_nearest(self):
        """... of which a table entry with the nearest row count lent its tile (mlctx_plan_tune_nearest); the others use the static rule."""
        f = L().mlctx_plan_tune_nearest
        f.argtypes = [vp]
        return int(f(self.h))

    def compute(self):
        check1(L().mlctx_compute(self.h), "mlctx_compute")

    def sync(self):
        check1(L().mlctx_sync(self.h), "mlctx_sync")

    def destroy(self):
        if self.h:
            L().mlctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def unet_params(model):
    u = UnetParams()
    check1(L().unet_params_get(model.encode(), ctypes.byref(u)), "unet_params_get")
    return u


class Unet:
    """unet_denoise_init_n / unet_denoise_run_n (src/unet.c:336-498) with a batch dimension."""

    def __init__(self, model, lw, lh, n_batch, stream=None, flags=0, seed=1234, synth=True, stream_weights_mib=0):
        self.P = unet_params(model)
        self.ctx = MLCtx(stream, flags)
        if stream_weights_mib:          # the reference's --unet-split: weights in pinned host memory, three device slabs of this size
            f = L().mlctx_set_weight_streaming
            f.argtypes = [vp, ctypes.c_size_t]
            check1(f(self.ctx.h, int(stream_weights_mib) << 20), "mlctx_set_weight_streaming")
        self.S = UnetState()
        check1(L().unet_denoise_init_n(ctypes.byref(self.S), self.ctx.h, ctypes.byref(self.P), lw, lh, n_batch), "unet_denoise_init_n")
        check1(L().unet_denoise_build(ctypes.byref(self.S)), "unet_denoise_build")
        if synth:
            self.ctx.params_synth(seed)
        self.lw, self.lh, self.n = lw, lh, n_batch

    def run(self, x, cond, label, sigma):
        x = np.ascontiguousarray(x, np.float32)
        cond = np.ascontiguousarray(cond, np.float32)
        sigma = np.ascontiguousarray(sigma, np.float32)
        lab = np.ascontiguousarray(label, np.float32) if label is not None else None
        dx = np.empty_like(x)
        check1(L().unet_denoise_run_n(ctypes.byref(self.S), fptr(x), fptr(cond), fptr(lab), fptr(sigma), fptr(dx)), "unet_denoise_run_n")
        return dx


def _proto2():
    l = L()
    if getattr(l, "_proto2_done", False):
        return l
    l.vae_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(VaeParams)]
    l.sdvae_decode_init.argtypes = [vp, ctypes.POINTER(VaeParams), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(vp)]
    l.sdvae_decode_build.argtypes = [vp, ctypes.POINTER(VaeParams), vp]
    l.sdvae_decode_run.argtypes = [vp, vp, FP, FP]
    l.sdtae_decode_init.argtypes = [vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(vp)]
    l.sdtae_decode_build.argtypes = [vp, vp]
    l.sdtae_decode_run.argtypes = [vp, vp, FP, FP]
    l.clip_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(ClipParams)]
    l.clip_text_encode.argtypes = [vp, ctypes.POINTER(ClipParams), ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint,
                                   ctypes.POINTER(ctypes.c_int32), FP, FP, c_int, ctypes.c_bool, c_u64]
    l.rng_philox_randn.argtypes = [vp, ctypes.c_uint, FP]
    l.dnsamp_schedule.argtypes = [ctypes.POINTER(UnetParams), c_int, c_int, c_f, c_f, FP]
    l.dnsamp_ancestral.argtypes = [c_f, c_f, c_f, FP, FP]
    l.mlis_amd_create.restype = vp
    l.mlis_amd_create.argtypes = [ctypes.POINTER(AmdConfig), vp]
    l.mlis_amd_destroy.argtypes = [vp]
    l.mlis_amd_set_cond.argtypes = [vp, FP, FP, FP, FP]
    l.mlis_amd_set_cond_device.argtypes = [vp, vp, vp, vp, vp]
    l.mlis_amd_generate.argtypes = [vp, ctypes.POINTER(c_u64), FP, FP]
    l.mlis_amd_denoise.argtypes = [vp, ctypes.POINTER(c_u64)]
    l.mlis_amd_decode.argtypes = [vp]
    l.mlis_amd_latent_device.restype = vp
    l.mlis_amd_latent_device.argtypes = [vp]
    l.mlis_amd_image_device.restype = vp
    l.mlis_amd_image_device.argtypes = [vp]
    l.mlis_amd_info.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int),
                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    l.mlis_amd_unet_ctx.restype = vp
    l.mlis_amd_unet_ctx.argtypes = [vp]
    l.mlis_amd_decoder_ctx.restype = vp
    l.mlis_amd_decoder_ctx.argtypes = [vp]
    l.mlis_amd_last_unet_ms.restype = c_f
    l.mlis_amd_last_unet_ms.argtypes = [vp]
    l.mlis_amd_last_nfe.argtypes = [vp]
    l.mlis_amd_last_n_step.argtypes = [vp]
    l.mlis_amd_sync.argtypes = [vp]
    l.mlis_amd_set_vae_tile.argtypes = [vp, c_int]
    l.mlis_amd_seed.argtypes = [vp, ctypes.POINTER(c_u64)]
    l.mlis_amd_set_init_latent.argtypes = [vp, FP]
    l.mlis_amd_set_lmask.argtypes = [vp, FP]
    l.mlis_amd_encode.argtypes = [vp, FP, c_int]
    l.mlis_amd_encoder_ctx.restype = vp
    l.mlis_amd_encoder_ctx.argtypes = [vp]
    l.mlis_amd_set_callback.argtypes = [vp, vp, vp]
    l._proto2_done = True
    return l


class Rng(ctypes.Structure):
    _fields_ = [("seed", c_u64), ("offset", ctypes.c_uint32)]


def randn(seed, offset, n):
    """rng_philox_randn of the host library (src/ccommon/rng_philox.c restated)."""
    r = Rng(seed, offset)
    out = np.empty(n, np.float32)
    _proto2().rng_philox_randn(ctypes.byref(r), n, fptr(out))
    return out, r.offset


def schedule(model, n_step, sched=1, f_t_ini=1.0, f_t_end=0.0):
    P = unet_params(model)
    sig = np.zeros(n_step + 2, np.float32)
    n = _proto2().dnsamp_schedule(ctypes.byref(P), n_step, sched, f_t_ini, f_t_end, fptr(sig))
    check1(n, "dnsamp_schedule")
    return sig[:n + 1]


class Decoder:
    """sdvae_decode / sdtae_decode (src/vae.c:318-411, src/tae.c:117-136), batched, no tiling."""

    def __init__(self, model, lw, lh, n_batch, tae=False, stream=None, seed=1234, flags=0):
        l = _proto2()
        self.ctx = MLCtx(stream, flags)
        self.tae, self.lw, self.lh, self.n = tae, lw, lh, n_batch
        self.t_lat = vp()
        if tae:
            check1(l.sdtae_decode_init(self.ctx.h, lw, lh, n_batch, ctypes.byref(self.t_lat)), "sdtae_decode_init")
            check1(l.sdtae_decode_build(self.ctx.h, self.t_lat), "sdtae_decode_build")
        else:
            self.P = VaeParams()
            check1(l.vae_params_get(model.encode(), ctypes.byref(self.P)), "vae_params_get")
            check1(l.sdvae_decode_init(self.ctx.h, ctypes.byref(self.P), lw, lh, n_batch, ctypes.byref(self.t_lat)), "sdvae_decode_init")
            check1(l.sdvae_decode_build(self.ctx.h, ctypes.byref(self.P), self.t_lat), "sdvae_decode_build")
        self.ctx.params_synth(seed)

    def run(self, latent):
        latent = np.ascontiguousarray(latent, np.float32)
        img = np.empty((self.n, 3, self.lh * 8, self.lw * 8), np.float32)
        f = _proto2().sdtae_decode_run if self.tae else _proto2().sdvae_decode_run
        check1(f(self.ctx.h, self.t_lat, fptr(latent), fptr(img)), "decode_run")
        return img


def clip_text_encode(model, prefix, toks, want_embed=True, want_feat=False, clip_skip=1, norm=True, seed=1234, stream=None):
    """clip_text_encode (src/clip.c:439-488) for a batch of prompts; toks int32 [n_prompt][n_tok]."""
    l = _proto2()
    P = ClipParams()
    check1(l.clip_params_get(model.encode(), ctypes.byref(P)), "clip_params_get")
    toks = np.ascontiguousarray(toks, np.int32)
    n_prompt, n_tok = toks.shape
    embed = np.empty((n_prompt, P.n_token, P.d_embed), np.float32) if want_embed else None
    feat = np.empty((n_prompt, P.d_embed), np.float32) if want_feat else None
    ctx = MLCtx(stream)
    check1(l.clip_text_encode(ctx.h, ctypes.byref(P), prefix.encode(), n_prompt, n_tok,
                              toks.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), fptr(embed), fptr(feat), clip_skip, norm, seed),
           "clip_text_encode")
    ctx.destroy()
    return embed, feat


class Generator:
    """mlis_amd_* generation driver (mlis_generate slice, src/mlimgsynth.c:1634-1773) for one GPU."""

    METHODS = {"euler": 1, "heun": 2, "taylor3": 3, "dpmpp2m": 4, "dpmpp2s": 5}

    def __init__(self, model, width, height, n_batch, n_step=20, cfg_scale=7.0, s_ancestral=1.0, sched=1, use_tae=False,
                 use_hipgraph=False, weight_seed=1234, stream=None, method="euler", s_noise=0.0, f_t_ini=1.0, f_t_end=0.0,
                 defer_weights=False, unet_split=0):
        l = _proto2()
        self.cfg = AmdConfig(model.encode(), width, height, n_batch, n_step, cfg_scale, s_ancestral, sched, int(use_tae),
                             int(use_hipgraph), weight_seed, self.METHODS.get(method, method), s_noise, f_t_ini, f_t_end,
                             int(defer_weights), int(unet_split))
        self.h = l.mlis_amd_create(ctypes.byref(self.cfg), vp(stream))
        if not self.h:
            from ._lib import MlsdError, last_error
            raise MlsdError("mlis_amd_create failed: " + last_error())
        self.model, self.w, self.h_px, self.B = model, width, height, n_batch
        self.P = unet_params(model)

    def set_cond(self, cond, label=None, uncond=None, unlabel=None):
        a = [np.ascontiguousarray(x, np.float32) if x is not None else None for x in (cond, label, uncond, unlabel)]
        check1(_proto2().mlis_amd_set_cond(self.h, *[fptr(x) for x in a]), "mlis_amd_set_cond")

    def set_cond_device(self, cond, label=None, uncond=None, unlabel=None):
        check1(_proto2().mlis_amd_set_cond_device(self.h, vp(cond), vp(label), vp(uncond), vp(unlabel)), "mlis_amd_set_cond_device")

    def set_init_latent(self, latent):
        a = np.ascontiguousarray(latent, np.float32) if latent is not None else None
        check1(_proto2().mlis_amd_set_init_latent(self.h, fptr(a)), "mlis_amd_set_init_latent")

    def set_lmask(self, lmask):
        a = np.ascontiguousarray(lmask, np.float32) if lmask is not None else None
        check1(_proto2().mlis_amd_set_lmask(self.h, fptr(a)), "mlis_amd_set_lmask")

    def seed(self, seeds):
        check1(_proto2().mlis_amd_seed(self.h, (c_u64 * self.B)(*[int(s) for s in seeds])), "mlis_amd_seed")

    def encode(self, images, sample=True):
        """mlis_image_encode: images [B][3][H][W] in [0,1] -> the resident latent (returned as numpy too)"""
        a = np.ascontiguousarray(images, np.float32)
        check1(_proto2().mlis_amd_encode(self.h, fptr(a), int(sample)), "mlis_amd_encode")
        from ._lib import lib
        out = np.empty((self.B, 4, self.h_px // 8, self.w // 8), np.float32)
        lib().mlsd_memcpy(out.ctypes.data_as(vp), vp(self.latent_ptr()), ctypes.c_size_t(out.nbytes), 1, None)
        lib().mlsd_device_sync()
        return out

    def last_n_step(self):
        return _proto2().mlis_amd_last_n_step(self.h)

    def generate(self, seeds, want_latents=True, want_images=True):
        seeds = (c_u64 * self.B)(*[int(s) for s in seeds]) if seeds is not None else None
        lat = np.empty((self.B, 4, self.h_px // 8, self.w // 8), np.float32) if want_latents else None
        img = np.empty((self.B, 3, self.h_px, self.w), np.float32) if want_images else None
        check1(_proto2().mlis_amd_generate(self.h, seeds, fptr(lat), fptr(img)), "mlis_amd_generate")
        return lat, img

    def denoise(self, seeds):
        seeds = (c_u64 * self.B)(*[int(s) for s in seeds])
        check1(_proto2().mlis_amd_denoise(self.h, seeds), "mlis_amd_denoise")

    def decode(self):
        check1(_proto2().mlis_amd_decode(self.h), "mlis_amd_decode")
        check1(_proto2().mlis_amd_sync(self.h), "mlis_amd_sync")

    def set_vae_tile(self, tile_px):
        check1(_proto2().mlis_amd_set_vae_tile(self.h, int(tile_px)), "mlis_amd_set_vae_tile")

    def image(self):
        from ._lib import lib
        out = np.empty((self.B, 3, self.h_px, self.w), np.float32)
        lib().mlsd_memcpy(out.ctypes.data_as(vp), vp(self.image_ptr()), ctypes.c_size_t(out.nbytes), 1, None)
        lib().mlsd_device_sync()
        return out

    def latent_ptr(self):
        return _proto2().mlis_amd_latent_device(self.h)

    def image_ptr(self):
        return _proto2().mlis_amd_image_device(self.h)

    def info(self):
        uf, df, ops, mp, mc = ctypes.c_double(), ctypes.c_double(), c_int(), ctypes.c_size_t(), ctypes.c_size_t()
        _proto2().mlis_amd_info(self.h, ctypes.byref(uf), ctypes.byref(df), ctypes.byref(ops), ctypes.byref(mp), ctypes.byref(mc))
        return dict(unet_flops=uf.value, decode_flops=df.value, unet_ops=ops.value, mem_params=mp.value, mem_compute=mc.value)

    def last_unet_ms(self):
        return _proto2().mlis_amd_last_unet_ms(self.h)

    def last_nfe(self):
        return _proto2().mlis_amd_last_nfe(self.h)

    def unet_ctx(self):
        c = MLCtx.__new__(MLCtx)
        c.h = _proto2().mlis_amd_unet_ctx(self.h)
        c.destroy = lambda: None
        return c

    def decoder_ctx(self):
        c = MLCtx.__new__(MLCtx)
        c.h = _proto2().mlis_amd_decoder_ctx(self.h)
        c.destroy = lambda: None
        return c

    def destroy(self):
        if self.h:
            _proto2().mlis_amd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
