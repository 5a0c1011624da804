"""ctypes mirror of include/mlsd_kernels.h (the op-level C-ABI shim into the HIP kernels)."""
import ctypes

from . import _lib
from ._lib import check, lib, vp

c_i64, c_int, c_f = ctypes.c_int64, ctypes.c_int, ctypes.c_float

ACT_NONE, ACT_SILU, ACT_GELU, ACT_GELU_QUICK, ACT_RELU, ACT_GEGLU = range(6)


class GemmArgs(ctypes.Structure):
    _fields_ = [("A", vp), ("lda", c_i64), ("conv", c_int), ("n_img", c_int), ("H", c_int), ("W", c_int),
                ("Cin", c_int), ("OH", c_int), ("OW", c_int), ("KH", c_int), ("KW", c_int), ("stride", c_int),
                ("pad", c_int), ("upsample", c_int), ("W_", vp), ("ldb", c_i64), ("M", c_int), ("N", c_int),
                ("K", c_int), ("bias", vp), ("rowbias", vp), ("rows_per_batch", c_int), ("ldrb", c_i64),
                ("resid", vp), ("ldr", c_i64), ("act", c_int), ("C32", vp), ("ldc32", c_i64), ("C16", vp),
                ("ldc16", c_i64), ("bias_m", vp), ("act_after_resid", c_int), ("tile_variant", c_int),
                ("ksplit", c_int), ("ws", vp), ("ws_bytes", ctypes.c_size_t), ("colstats", vp), ("colstats_rows", c_int), ("sk_flags", vp),
                ("ln_y16", vp), ("ldln", ctypes.c_int64), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", ctypes.c_float), ("ln_ws", vp), ("ln_cnt", vp), ("ln_slot", c_int),
                ("gn_y16", vp), ("gn_ldy", ctypes.c_int64), ("gn_gamma", vp), ("gn_beta", vp), ("gn_eps", ctypes.c_float), ("gn_groups", c_int), ("gn_hw", c_int),
                ("gn_silu", c_int),
                ("xa_k", vp), ("xa_ldk", c_i64), ("xa_vt", vp), ("xa_out", vp), ("xa_ldo", c_i64), ("xa_Tq", c_int), ("xa_Tk", c_int)]


class AttnArgs(ctypes.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp), ("ldq", c_i64), ("ldk", c_i64), ("ldv", c_i64),
                ("ldo", c_i64), ("bsq", c_i64), ("bsk", c_i64), ("bsv", c_i64), ("bso", c_i64), ("n_batch", c_int),
                ("n_head", c_int), ("d_head", c_int), ("Tq", c_int), ("Tk", c_int), ("causal", c_int)]


class GnArgs(ctypes.Structure):
    _fields_ = [("x1", vp), ("x2", vp), ("ld1", c_i64), ("ld2", c_i64), ("C1", c_int), ("C2", c_int),
                ("n_img", c_int), ("HW", c_int), ("n_grp", c_int), ("eps", c_f), ("gamma", vp), ("beta", vp),
                ("silu", c_int), ("y16", vp), ("raw16", vp), ("ws", vp), ("cs1", vp), ("cs2", vp), ("rb_rows1", c_int), ("rb_rows2", c_int)]


def gemm(args, stream=None):
    check(lib().mlsd_gemm(ctypes.byref(args), vp(stream)), "mlsd_gemm")


def gemm_splitk_ws_bytes(m, n, ksplit):
    f = lib().mlsd_gemm_splitk_ws_bytes
    f.restype = ctypes.c_size_t
    return f(m, n, ksplit)


def gemm_variant(args):
    f = lib().mlsd_gemm_variant
    f.restype = ctypes.c_char_p
    return f(ctypes.byref(args)).decode()


def attention(args, stream=None):
    check(lib().mlsd_attention(ctypes.byref(args), vp(stream)), "mlsd_attention")


def groupnorm(args, stream=None):
    check(lib().mlsd_groupnorm(ctypes.byref(args), vp(stream)), "mlsd_groupnorm")


def groupnorm_ws_bytes(n_img, hw, n_grp):
    f = lib().mlsd_groupnorm_ws_bytes
    f.restype = ctypes.c_size_t
    return f(n_img, hw, n_grp)


def layernorm(x, ldx, rows, d, eps, gamma, beta, y16, y32=None, stream=None):
    check(lib().mlsd_layernorm(vp(x), c_i64(ldx), rows, d, c_f(eps), vp(gamma), vp(beta), vp(y16), vp(y32), vp(stream)),
          "mlsd_layernorm")


def sync():
    check(lib().mlsd_device_sync(), "sync")
