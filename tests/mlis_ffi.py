"""ctypes binding of the public libmlimgsynth API built from a PROTOTYPE TABLE (data): the symbols, restype and argtypes
that the reference's own FFI declares (python/mlimgsynth.py:182-208) plus the remaining functions of
include/mlimgsynth.h:414-571.  The library path is the only thing that differs from the reference binding
(MLIS_LIB_PATH -> libmlimgsynth_amd.so)."""
import ctypes as C

vp, ci, cf, cs = C.c_void_p, C.c_int, C.c_float, C.c_char_p


class Image(C.Structure):        # MLIS_Image, mlimgsynth.h:366-373
    _fields_ = [("d", C.POINTER(C.c_uint8)), ("sz", C.c_size_t), ("w", C.c_uint), ("h", C.c_uint), ("c", C.c_uint), ("flags", ci)]


class Tensor(C.Structure):       # MLIS_Tensor, mlimgsynth.h:409-413
    _fields_ = [("d", C.POINTER(cf)), ("n", ci * 4), ("flags", ci)]


class Progress(C.Structure):     # MLIS_Progress, mlimgsynth.h:377-384
    _fields_ = [("stage", ci), ("step", ci), ("step_end", ci), ("nfe", ci), ("step_time", C.c_double), ("time", C.c_double)]


class ErrorInfo(C.Structure):
    _fields_ = [("code", ci), ("desc", cs)]


CALLBACK = C.CFUNCTYPE(ci, vp, vp, C.POINTER(Progress))
ERRHANDLER = C.CFUNCTYPE(None, vp, vp, C.POINTER(ErrorInfo))

# (symbol, restype, argtypes) -- first block: exactly python/mlimgsynth.py:182-208
PROTOTYPES = [
    ("mlis_ctx_create_i", vp, [ci]),
    ("mlis_ctx_destroy", None, [C.POINTER(vp)]),
    ("mlis_errstr_get", cs, [vp]),
    ("mlis_option_set", ci, [vp, ci]),                       # variadic: extra arguments by ctypes' default promotions
    ("mlis_option_set_str", ci, [vp, cs, cs]),
    ("mlis_generate", ci, [vp]),
    ("mlis_image_get", C.POINTER(Image), [vp, ci]),
    ("mlis_infotext_get", cs, [vp, ci]),
    ("mlis_setup", ci, [vp]),
    ("mlis_tensor_get", C.POINTER(Tensor), [vp, ci]),
    ("mlis_clip_text_encode", ci, [vp, cs, C.POINTER(Tensor), C.POINTER(Tensor), ci, ci]),
    ("mlis_tensor_similarity", cf, [C.POINTER(Tensor), C.POINTER(Tensor)]),
    # the rest of include/mlimgsynth.h:414-571
    ("mlis_option_get", ci, [vp, ci]),
    ("mlis_backend_info_get", vp, [vp, C.c_uint, ci]),
    ("mlis_stage_str", cs, [ci]), ("mlis_stage_desc", cs, [ci]), ("mlis_stage_fromz", ci, [cs]),
    ("mlis_method_str", cs, [ci]), ("mlis_method_fromz", ci, [cs]),
    ("mlis_sched_str", cs, [ci]), ("mlis_sched_fromz", ci, [cs]),
    ("mlis_loglvl_str", cs, [ci]), ("mlis_loglvl_fromz", ci, [cs]),
    ("mlis_model_type_str", cs, [ci]), ("mlis_model_type_desc", cs, [ci]), ("mlis_model_type_fromz", ci, [cs]),
    ("mlis_option_str", cs, [ci]), ("mlis_option_fromz", ci, [cs]),
    ("mlis_image_encode", ci, [vp, C.POINTER(Tensor), C.POINTER(Tensor), ci]),
    ("mlis_image_decode", ci, [vp, C.POINTER(Tensor), C.POINTER(Tensor), ci]),
    ("mlis_mask_encode", ci, [vp, C.POINTER(Tensor), C.POINTER(Tensor), ci]),
    ("mlis_text_tokenize", ci, [vp, cs, C.POINTER(C.POINTER(C.c_int32)), ci]),
    ("mlis_tensor_free", None, [C.POINTER(Tensor)]),
    ("mlis_tensor_count", C.c_size_t, [C.POINTER(Tensor)]),
    ("mlis_tensor_resize", None, [C.POINTER(Tensor), ci, ci, ci, ci]),
    ("mlis_tensor_resize_like", None, [C.POINTER(Tensor), C.POINTER(Tensor)]),
    ("mlis_tensor_copy", None, [C.POINTER(Tensor), C.POINTER(Tensor)]),
    # additions of this library
    ("mlis_amd_prompt_tokens_set", ci, [vp, C.POINTER(C.c_int32), C.POINTER(cf), ci, ci]),
]
MLIS_VERSION = 0x000402
OPT = dict(BACKEND=1, MODEL=2, TAE=3, PROMPT=7, NPROMPT=8, IMAGE_DIM=9, BATCH_SIZE=10, CLIP_SKIP=11, CFG_SCALE=12, METHOD=13, SCHEDULER=14,
           STEPS=15, F_T_INI=16, F_T_END=17, S_NOISE=18, S_ANCESTRAL=19, IMAGE=20, IMAGE_MASK=21, NO_DECODE=22, TENSOR_USE_FLAGS=23,
           SEED=24, VAE_TILE=25, AUX_DIR=29, CALLBACK=30, ERROR_HANDLER=31, MODEL_TYPE=33, WEIGHT_TYPE=34, NO_PROMPT_PARSE=35)
TENSOR = dict(IMAGE=1, MASK=2, LATENT=3, LMASK=4, COND=5, LABEL=6, NCOND=7, NLABEL=8, TMP=0x100)
TUF = dict(IMAGE=1, MASK=2, LATENT=4, LMASK=8, CONDITIONING=16)


def bind(path):
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    missing = []
    for name, res, args in PROTOTYPES:
        if not hasattr(lib, name):
            missing.append(name)
            continue
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    assert not missing, missing
    return lib


def tensor_np(t):
    import numpy as np
    n = [int(t.n[i]) for i in range(4)]
    cnt = n[0] * n[1] * n[2] * n[3]
    return np.ctypeslib.as_array(t.d, shape=(cnt,)).copy().reshape(n[3], n[2], n[1], n[0])


class Mlis:
    """what python/mlimgsynth.py's MLImgSynth class does, on the prototype table"""

    def __init__(self, lib):
        self.lib = lib
        self.ctx = vp(lib.mlis_ctx_create_i(MLIS_VERSION))
        assert self.ctx.value

    def close(self):
        if self.ctx and self.ctx.value:
            self.lib.mlis_ctx_destroy(C.byref(self.ctx))

    def err(self):
        e = self.lib.mlis_errstr_get(self.ctx)
        return e.decode() if e else ""

    def set(self, name, *args):
        """string form like the reference wrapper: arguments joined with ','"""
        r = self.lib.mlis_option_set_str(self.ctx, name.encode(), ",".join(str(a) for a in args).encode())
        if r < 0:
            raise RuntimeError(f"option {name}: {self.err()}")
        return r

    def generate(self):
        r = self.lib.mlis_generate(self.ctx)
        if r < 0:
            raise RuntimeError(f"generate ({r}): {self.err()}")

    def tensor(self, tid):
        return tensor_np(self.lib.mlis_tensor_get(self.ctx, tid).contents)

    def image(self, idx=0):
        import numpy as np
        p = self.lib.mlis_image_get(self.ctx, idx)
        if not p:
            raise RuntimeError(self.err())
        im = p.contents
        return np.ctypeslib.as_array(im.d, shape=(im.h, im.w, im.c)).copy()

    def tokens(self, toks, weights=None, negative=False):
        import numpy as np
        t = np.ascontiguousarray(toks, np.int32)
        w = np.ascontiguousarray(weights, np.float32) if weights is not None else None
        r = self.lib.mlis_amd_prompt_tokens_set(self.ctx, t.ctypes.data_as(C.POINTER(C.c_int32)),
                                                w.ctypes.data_as(C.POINTER(cf)) if w is not None else None, t.size, int(negative))
        assert r > 0, self.err()
