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
                ("weight_seed", c_u64)]


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
        l.mlctx_compute.argtypes = [vp]
        l.mlctx_sync.argtypes = [vp]
        l.unet_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(UnetParams)]
        l.unet_denoise_init.argtypes = [ctypes.POINTER(UnetState), vp, ctypes.POINTER(UnetParams), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
        l.unet_denoise_build.argtypes = [ctypes.POINTER(UnetState)]
        l.unet_denoise_run.argtypes = [ctypes.POINTER(UnetState), FP, FP, FP, FP, FP]
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

    def profile_ops(self):
        n = self.info().n_ops
        ms = np.zeros(n, np.float32)
        check1(L().mlctx_profile_ops(self.h, fptr(ms), n), "mlctx_profile_ops")
        return ms

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
    """unet_denoise_init / unet_denoise_run (src/unet.c:336-498) with a batch dimension."""

    def __init__(self, model, lw, lh, n_batch, stream=None, flags=0, seed=1234, synth=True):
        self.P = unet_params(model)
        self.ctx = MLCtx(stream, flags)
        self.S = UnetState()
        check1(L().unet_denoise_init(ctypes.byref(self.S), self.ctx.h, ctypes.byref(self.P), lw, lh, n_batch), "unet_denoise_init")
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
        check1(L().unet_denoise_run(ctypes.byref(self.S), fptr(x), fptr(cond), fptr(lab), fptr(sigma), fptr(dx)), "unet_denoise_run")
        return dx
