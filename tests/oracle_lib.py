"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY: the CPU restatement
of the reference path used as the checker).  Never imported by the product package."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_RNG_SO = os.path.join(ROOT, "oracle", "_ref", "librng_ref.so")

c_i64, c_f, c_vp, c_int = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_int
FP = ctypes.POINTER(ctypes.c_float)


class OT(ctypes.Structure):
    _fields_ = [("ne", c_i64 * 4), ("d", FP)]


class OParam(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("type", c_int), ("ne", c_i64 * 4), ("d", FP)]


class UnetParams(ctypes.Structure):
    _fields_ = [("n_ch_in", c_int), ("n_ch_out", c_int), ("n_res_blk", c_int), ("attn_res", c_int * 4),
                ("ch_mult", c_int * 5), ("transf_depth", c_int * 5), ("n_te", c_int), ("n_head", c_int),
                ("d_head", c_int), ("n_ctx", c_int), ("n_ch", c_int), ("ch_adm_in", c_int),
                ("clip_norm", c_int), ("cond_label", c_int), ("uncond_empty_zero", c_int), ("vparam", c_int),
                ("n_step_train", c_int), ("sigma_min", c_f), ("sigma_max", c_f)]


class VaeParams(ctypes.Structure):
    _fields_ = [("ch_x", c_int), ("ch_z", c_int), ("ch", c_int), ("n_res", c_int), ("n_res_blk", c_int),
                ("ch_mult", c_int * 5), ("d_embed", c_int), ("f_down", c_int), ("scale_factor", c_f)]


class ClipParams(ctypes.Structure):
    _fields_ = [("n_vocab", c_int), ("n_token", c_int), ("d_embed", c_int), ("n_interm", c_int),
                ("n_head", c_int), ("n_layer", c_int), ("tok_start", c_int), ("tok_end", c_int), ("tok_pad", c_int)]


class SampleOpts(ctypes.Structure):
    _fields_ = [("method", c_int), ("sched", c_int), ("n_step", c_int), ("cfg_scale", c_f), ("s_ancestral", c_f),
                ("s_noise", c_f), ("f_t_ini", c_f), ("f_t_end", c_f)]


class Rng(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("offset", ctypes.c_uint32)]


_L = None


def host_threads(cap=64):
    """OpenMP threads for the oracle: the CPUs this process may really use (scheduler affinity, cgroup quota), capped.  The GPU boxes show 256 logical CPUs and grant 16
    (cpu.max): the oracle's SGEMM ran at 2.9 TFLOP/s on 16 threads, 1.3 on 128 and 0.06 on 256 (profiles/r5_cpu_probe.txt).  Same rule as bench.py's host_threads()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, cap))


def L():
    global _L
    if _L is None:
        if not os.path.exists(ORACLE_SO):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        l = ctypes.CDLL(ORACLE_SO)
        l.orc_set_threads(host_threads())          # (never OpenMP's default of one thread per logical CPU: see host_threads)
        OTP = ctypes.POINTER(OT)
        OPP = ctypes.POINTER(OParam)
        l.ot_new.restype = OTP
        l.ot_new.argtypes = [c_i64] * 4
        l.ot_from.restype = OTP
        l.ot_from.argtypes = [FP] + [c_i64] * 4
        l.ot_free.argtypes = [OTP]
        l.ot_nel.restype = c_i64
        l.ot_nel.argtypes = [OTP]
        l.orc_round_f16.argtypes = [FP, c_i64]
        l.orc_sgemm_nt.argtypes = [c_i64, c_i64, c_i64, FP, c_i64, FP, c_i64, FP, c_i64]
        l.orc_params_new.restype = c_vp
        l.orc_params_new.argtypes = [ctypes.c_uint64]
        l.orc_params_free.argtypes = [c_vp]
        l.orc_params_set.argtypes = [c_vp, ctypes.c_char_p, c_int] + [c_i64] * 4 + [FP]
        l.orc_params_get.restype = OPP
        l.orc_params_get.argtypes = [c_vp, ctypes.c_char_p, c_int] + [c_i64] * 4
        l.orc_params_count.argtypes = [c_vp]
        l.orc_params_at.restype = OPP
        l.orc_params_at.argtypes = [c_vp, c_int]
        l.orc_synth_fill.argtypes = [FP, c_i64, ctypes.c_uint64, ctypes.c_char_p, c_f, c_f, c_int]
        l.orc_synth_rule.argtypes = [ctypes.c_char_p, c_int, ctypes.POINTER(c_i64 * 4), FP, FP]
        l.orc_linear.restype = OTP
        l.orc_linear.argtypes = [OTP, OPP, OPP]
        l.orc_conv2d.restype = OTP
        l.orc_conv2d.argtypes = [OTP, OPP, OPP, c_int, c_int]
        l.orc_group_norm.restype = OTP
        l.orc_group_norm.argtypes = [OTP, c_int, c_f, OPP, OPP]
        l.orc_layer_norm.restype = OTP
        l.orc_layer_norm.argtypes = [OTP, c_f, OPP, OPP]
        l.orc_attention.restype = OTP
        l.orc_attention.argtypes = [OTP, OTP, OTP, c_int, c_int]
        for f in ("orc_silu", "orc_gelu", "orc_gelu_quick", "orc_relu"):
            getattr(l, f).argtypes = [OTP]
        l.orc_upscale2.restype = OTP
        l.orc_upscale2.argtypes = [OTP]
        l.orc_pad_end.restype = OTP
        l.orc_pad_end.argtypes = [OTP, c_int, c_int]
        l.orc_concat_ch.restype = OTP
        l.orc_concat_ch.argtypes = [OTP, OTP]
        l.orc_timestep_embedding.argtypes = [FP, c_int, c_int, c_f, FP]
        l.orc_unet_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(UnetParams)]
        l.orc_vae_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(VaeParams)]
        l.orc_clip_params_get.argtypes = [ctypes.c_char_p, ctypes.POINTER(ClipParams)]
        l.orc_unet_graph.restype = OTP
        l.orc_unet_graph.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(UnetParams), OTP, c_f, OTP, OTP]
        l.orc_unet_denoise_run.restype = OTP
        l.orc_unet_denoise_run.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(UnetParams), OTP, OTP, OTP, c_f]
        l.orc_vae_decode.restype = OTP
        l.orc_vae_decode.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(VaeParams), OTP]
        l.orc_tae_decode.restype = OTP
        l.orc_tae_decode.argtypes = [c_vp, ctypes.c_char_p, OTP]
        l.orc_clip_text_encode.restype = OTP
        l.orc_clip_text_encode.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(ClipParams),
                                           ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_int, c_int]
        l.orc_log_sigmas.argtypes = [FP]
        l.orc_sigma_to_t.restype = c_f
        l.orc_sigma_to_t.argtypes = [c_f]
        l.orc_t_to_sigma.restype = c_f
        l.orc_t_to_sigma.argtypes = [c_f]
        l.orc_schedule.argtypes = [c_int, c_int, c_f, c_f, FP]
        l.orc_ancestral.argtypes = [c_f, c_f, c_f, FP, FP]
        l.orc_rng_randn.argtypes = [ctypes.POINTER(Rng), ctypes.c_uint, FP]
        l.orc_philox_raw.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint32)]
        l.orc_generate_latent.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(UnetParams), c_int, c_int,
                                          OTP, OTP, OTP, OTP, c_f, c_int, c_f, ctypes.c_uint64, c_int, FP,
                                          ctypes.POINTER(ctypes.c_double)]
        l.orc_set_threads.argtypes = [c_int]
        l.orc_set_act_rounding.argtypes = [c_int]
        l.orc_vae_encode_moments.restype = OTP
        l.orc_vae_encode_moments.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(VaeParams), OTP]
        l.orc_vae_decode_tiled.restype = OTP
        l.orc_vae_decode_tiled.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(VaeParams), OTP, c_int]
        l.orc_vae_encode_moments_tiled.restype = OTP
        l.orc_vae_encode_moments_tiled.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(VaeParams), OTP, c_int]
        l.orc_latent_sample.restype = OTP
        l.orc_latent_sample.argtypes = [OTP, ctypes.POINTER(VaeParams), FP]
        l.orc_tae_encode.restype = OTP
        l.orc_tae_encode.argtypes = [c_vp, ctypes.c_char_p, OTP]
        l.orc_mask_downsize.argtypes = [FP, c_int, c_int, c_int, FP]
        l.orc_sample_ex.argtypes = [c_vp, ctypes.c_char_p, ctypes.POINTER(UnetParams), c_int, c_int, OTP, OTP, OTP, OTP,
                                    ctypes.POINTER(SampleOpts), ctypes.c_uint64, ctypes.c_uint32, FP, FP, FP]
        _L = l
    return _L


def fptr(a):
    return a.ctypes.data_as(FP)


def to_ot(arr):
    """numpy array with shape in ggml order REVERSED (i.e. numpy shape (n3,n2,n1,n0) trailing dims) -> OT*.
    Pass arrays shaped like torch: e.g. NCHW (N,C,H,W) -> ne = [W,H,C,N]."""
    a = np.ascontiguousarray(arr, dtype=np.float32)
    shp = list(a.shape)[::-1] + [1] * (4 - a.ndim)
    return L().ot_from(fptr(a), *shp)


def from_ot(t, free=True):
    ne = [int(t.contents.ne[i]) for i in range(4)]
    n = ne[0] * ne[1] * ne[2] * ne[3]
    a = np.ctypeslib.as_array(t.contents.d, shape=(n,)).copy().reshape(ne[3], ne[2], ne[1], ne[0])
    if free:
        L().ot_free(t)
    return a


class Params:
    def __init__(self, seed=1234):
        self.h = L().orc_params_new(seed)

    def set(self, name, arr, f16=False):
        """arr in torch-like shape (outermost first); stored with ne reversed."""
        a = np.ascontiguousarray(arr, dtype=np.float32)
        shp = list(a.shape)[::-1] + [1] * (4 - a.ndim)
        r = L().orc_params_set(self.h, name.encode(), 1 if f16 else 0, *shp, fptr(a))
        assert r == 1, name
        return self.get(name, f16, a.shape)

    def get(self, name, f16, shape):
        shp = list(shape)[::-1] + [1] * (4 - len(shape))
        p = L().orc_params_get(self.h, name.encode(), 1 if f16 else 0, *shp)
        assert p, name
        return p

    def get_np(self, name, f16, shape):
        p = self.get(name, f16, shape)
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(p.contents.d, shape=(n,)).copy().reshape(shape)

    def names(self):
        out = []
        for i in range(L().orc_params_count(self.h)):
            p = L().orc_params_at(self.h, i).contents
            out.append((p.name.decode(), p.type, [int(p.ne[k]) for k in range(4)]))
        return out

    def free(self):
        if self.h:
            L().orc_params_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def unet_params(model):
    u = UnetParams()
    L().orc_unet_params_get(model.encode(), ctypes.byref(u))
    return u


def vae_params(model):
    v = VaeParams()
    L().orc_vae_params_get(model.encode(), ctypes.byref(v))
    return v


def clip_params(model):
    c = ClipParams()
    L().orc_clip_params_get(model.encode(), ctypes.byref(c))
    return c


def randn(seed, offset, n):
    r = Rng(seed, offset)
    out = np.empty(n, np.float32)
    L().orc_rng_randn(ctypes.byref(r), n, fptr(out))
    return out


def ref_randn(seed, offset, n):
    """The reference's own rng_philox_randn compiled from /root/reference (oracle/_ref)."""
    ref = ctypes.CDLL(REF_RNG_SO)
    ref.rng_philox_randn.argtypes = [ctypes.POINTER(Rng), ctypes.c_uint, FP]
    r = Rng(seed, offset)
    out = np.empty(n, np.float32)
    ref.rng_philox_randn(ctypes.byref(r), n, fptr(out))
    return out, r.offset
