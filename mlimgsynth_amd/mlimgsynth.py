"""Python interface of libmlimgsynth on MI355X -- the counterpart of the reference's python/mlimgsynth.py.

Same surface: the MLIS_* constants, MLIS_Image / MLIS_Tensor (with .similarity) and the class MLImgSynth with
option_set / setup / generate / image_get / infotext_get / errstr_get / clip_text_encode (python/mlimgsynth.py:213-296),
so a script written against the reference imports this module instead and runs unchanged:

    from mlimgsynth_amd.mlimgsynth import *        # was: from mlimgsynth import *

plus the functions the reference wrapper lists as "TODO: all functions": tensor_get, option_get, text_tokenize,
image_encode / image_decode / mask_encode, callbacks, and numpy views of images and tensors.

The library is found like the reference does (environment MLIS_LIB_PATH first), falling back to the in-tree build
mlimgsynth_amd/lib/libmlimgsynth_amd.so.  It is loaded on first use, not at import, and there is no CPU fallback.
"""
import ctypes
import os
import sys

MLIS_VERSION = 0x000402
MLIS_VERSION_STR = "0.4.2"


def _define(prefix, names, start=0, explicit=None):
    g = globals()
    for i, n in enumerate(names):
        g[prefix + n] = start + i
    for n, v in (explicit or {}).items():
        g[prefix + n] = v


# enumerations of include/mlimgsynth.h (values are ABI)
_define("MLIS_E_", [], explicit=dict(UNKNOWN=-1, VERSION=-2, UNK_OPT=-3, OPT_VALUE=-4, PROMPT_PARSE=-5, FILE_NOT_FOUND=-6, NAN=-7, IMAGE=-8))
_define("MLIS_STAGE_", ["IDLE", "COND_ENCODE", "IMAGE_ENCODE", "IMAGE_DECODE", "DENOISE"])
_define("MLIS_METHOD_", ["NONE", "EULER", "HEUN", "TAYLOR3", "DPMPP2M", "DPMPP2S"], explicit=dict(_LAST=5))
_define("MLIS_SCHED_", ["NONE", "UNIFORM", "KARRAS"], explicit=dict(_LAST=2))
_define("MLIS_LOGLVL_", [], explicit=dict(NONE=0, ERROR=10, WARNING=20, INFO=30, VERBOSE=40, DEBUG=50, MAX=255, _INCREASE=0x100 | 10, _DECREASE=0x200 | 10))
_define("MLIS_TENSOR_", ["IMAGE", "MASK", "LATENT", "LMASK", "COND", "LABEL", "NCOND", "NLABEL"], start=1, explicit=dict(TMP=0x100))
_define("MLIS_TUF_", [], explicit=dict(IMAGE=1, MASK=2, LATENT=4, LMASK=8, CONDITIONING=16))
_define("MLIS_MODEL_TYPE_", ["NONE", "SD1", "SD2", "SDXL"], explicit=dict(_LAST=3))
_define("MLIS_MODEL_", ["NONE", "UNET", "VAE", "TAE", "CLIP", "CLIP2"])
_define("MLIS_OPT_", ["NONE", "BACKEND", "MODEL", "TAE", "LORA_DIR", "LORA", "LORA_CLEAR", "PROMPT", "NPROMPT", "IMAGE_DIM", "BATCH_SIZE",
                      "CLIP_SKIP", "CFG_SCALE", "METHOD", "SCHEDULER", "STEPS", "F_T_INI", "F_T_END", "S_NOISE", "S_ANCESTRAL", "IMAGE",
                      "IMAGE_MASK", "NO_DECODE", "TENSOR_USE_FLAGS", "SEED", "VAE_TILE", "UNET_SPLIT", "THREADS", "DUMP_FLAGS", "AUX_DIR",
                      "CALLBACK", "ERROR_HANDLER", "LOG_LEVEL", "MODEL_TYPE", "WEIGHT_TYPE", "NO_PROMPT_PARSE"], explicit=dict(_LAST=35))
MLIS_CTEF_NO_NORM = 1


class MLIS_Image_C(ctypes.Structure):          # MLIS_Image, include/mlimgsynth.h:366-373
    _fields_ = [("d", ctypes.POINTER(ctypes.c_uint8)), ("sz", ctypes.c_size_t), ("w", ctypes.c_int), ("h", ctypes.c_int),
                ("c", ctypes.c_int), ("flags", ctypes.c_int)]


class MLIS_Tensor_C(ctypes.Structure):         # MLIS_Tensor, include/mlimgsynth.h:409-413
    _fields_ = [("d", ctypes.POINTER(ctypes.c_float)), ("n", ctypes.c_int * 4), ("flags", ctypes.c_int)]


class MLIS_Progress_C(ctypes.Structure):       # MLIS_Progress, include/mlimgsynth.h:377-384
    _fields_ = [("stage", ctypes.c_int), ("step", ctypes.c_int), ("step_end", ctypes.c_int), ("nfe", ctypes.c_int),
                ("step_time", ctypes.c_double), ("time", ctypes.c_double)]


MLIS_Callback = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(MLIS_Progress_C))

_V, _I, _F, _S = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_char_p
_TP, _IP = ctypes.POINTER(MLIS_Tensor_C), ctypes.POINTER(MLIS_Image_C)
_PROTOTYPES = {
    "mlis_ctx_create_i": (_V, [_I]), "mlis_ctx_destroy": (None, [ctypes.POINTER(_V)]), "mlis_errstr_get": (_S, [_V]),
    "mlis_option_set": (_I, [_V, _I]), "mlis_option_set_str": (_I, [_V, _S, _S]), "mlis_option_get": (_I, [_V, _I]),
    "mlis_setup": (_I, [_V]), "mlis_generate": (_I, [_V]), "mlis_image_get": (_IP, [_V, _I]), "mlis_infotext_get": (_S, [_V, _I]),
    "mlis_tensor_get": (_TP, [_V, _I]), "mlis_clip_text_encode": (_I, [_V, _S, _TP, _TP, _I, _I]),
    "mlis_tensor_similarity": (_F, [_TP, _TP]), "mlis_image_encode": (_I, [_V, _TP, _TP, _I]), "mlis_image_decode": (_I, [_V, _TP, _TP, _I]),
    "mlis_mask_encode": (_I, [_V, _TP, _TP, _I]), "mlis_text_tokenize": (_I, [_V, _S, ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), _I]),
    "mlis_tensor_resize": (None, [_TP, _I, _I, _I, _I]), "mlis_tensor_free": (None, [_TP]),
}

mlis_lib = None
mlis_lib_path = None


def _find_library():
    p = os.getenv("MLIS_LIB_PATH", None)
    if p:
        return p
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in (os.path.join(here, "lib", "libmlimgsynth_amd.so"), "libmlimgsynth_amd.so", os.path.join("lib", "libmlimgsynth_amd.so")):
        if os.path.exists(cand):
            return cand
    raise RuntimeError("libmlimgsynth_amd.so not found: build it (python -c 'import __graft_entry__ as g; g.build()') or set MLIS_LIB_PATH")


def load_library(path=None):
    """Load the shared library (once).  Raises instead of falling back to anything else."""
    global mlis_lib, mlis_lib_path
    if mlis_lib is not None and path in (None, mlis_lib_path):
        return mlis_lib
    mlis_lib_path = path or _find_library()
    lib = ctypes.CDLL(mlis_lib_path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in _PROTOTYPES.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    mlis_lib = lib
    return lib


class MLIS_Image:
    def __init__(self, cimg):
        self.data = ctypes.string_at(cimg.d, cimg.sz)  # bytes, RGB
        self.w, self.h, self.c = int(cimg.w), int(cimg.h), int(cimg.c)

    def numpy(self):
        import numpy as np
        return np.frombuffer(self.data, np.uint8).reshape(self.h, self.w, self.c)


class MLIS_Tensor:
    def __init__(self, cten):
        self.n = tuple(int(x) for x in cten.n)
        self.data = ctypes.string_at(cten.d, self.n[0] * self.n[1] * self.n[2] * self.n[3] * 4)  # bytes, float32

    def _c(self):
        buf = ctypes.cast(ctypes.c_char_p(self.data), ctypes.POINTER(ctypes.c_float))
        return MLIS_Tensor_C(buf, (ctypes.c_int * 4)(*self.n), 0)

    def similarity(self, other):
        a, b = self._c(), other._c()
        return float(load_library().mlis_tensor_similarity(ctypes.byref(a), ctypes.byref(b)))

    def numpy(self):
        import numpy as np
        return np.frombuffer(self.data, np.float32).reshape(self.n[3], self.n[2], self.n[1], self.n[0])


class MLImgSynth:
    def __init__(self, lib_path=None):
        self._lib = load_library(lib_path)
        self._ctx = self._lib.mlis_ctx_create_i(MLIS_VERSION)
        if not self._ctx:
            raise RuntimeError("Failed to create MLIS context")
        self._keep = []          # callback thunks must outlive the context

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.mlis_ctx_destroy(ctypes.byref(ctypes.c_void_p(self._ctx)))
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- options
    def option_set(self, option, *args):
        if isinstance(option, str):
            r = self._lib.mlis_option_set_str(self._ctx, option.encode("utf8"), ",".join(str(x) for x in args).encode("utf8"))
        elif isinstance(option, int):
            conv = []
            for a in args:             # C default argument promotions of the variadic mlis_option_set
                conv.append(ctypes.c_double(a) if isinstance(a, float) else ctypes.c_char_p(a.encode("utf8")) if isinstance(a, str) else a)
            self._keep.extend(c for c in conv if isinstance(c, ctypes.c_char_p))
            r = self._lib.mlis_option_set(self._ctx, option, *conv)
        else:
            raise RuntimeError("'option' must be str or int")
        if r < 0:
            raise RuntimeError("Failed to set option '%s': %s" % (option, self.errstr_get()))
        return r

    def option_get(self, option, out):
        """out: a ctypes object receiving the value (c_char_p for MODEL / PROMPT / NPROMPT, c_int for MODEL_TYPE: the four options
        mlis_option_get implements, as in the reference)."""
        r = self._lib.mlis_option_get(self._ctx, option, ctypes.byref(out))
        if r < 0:
            raise RuntimeError("Failed to get option %s: %s" % (option, self.errstr_get()))
        return r

    def callback_set(self, fn, user=None):
        """fn(progress) -> int (>= 0 continue, < 0 abort with that code); progress has stage / step / step_end / nfe / step_time / time."""
        thunk = MLIS_Callback(lambda ud, ctx, p: int(fn(p.contents) or 0))
        self._keep.append(thunk)
        if self._lib.mlis_option_set(self._ctx, MLIS_OPT_CALLBACK, thunk, ctypes.c_void_p(user)) < 0:   # noqa: F821
            raise RuntimeError("Failed to set the callback: %s" % self.errstr_get())

    # ---- life cycle
    def setup(self):
        "Set up the backend and model. Optional."
        if self._lib.mlis_setup(self._ctx) < 0:
            raise RuntimeError("Failed to setup: %s" % (self.errstr_get()))

    def generate(self):
        "Generate images."
        r = self._lib.mlis_generate(self._ctx)
        if r < 0:
            raise RuntimeError("Failed to generate image: %s" % (self.errstr_get()))
        return r

    def image_get(self, idx=0):
        "Get generated images data."
        p = self._lib.mlis_image_get(self._ctx, idx)
        if not p:
            raise RuntimeError("Failed to get image %d" % idx)
        return MLIS_Image(p.contents)

    def tensor_get(self, tensor_id):
        "Copy of one of the library's tensors (MLIS_TENSOR_*)."
        p = self._lib.mlis_tensor_get(self._ctx, tensor_id)
        if not p or not p.contents.d:
            raise RuntimeError("Failed to get tensor %d" % tensor_id)
        return MLIS_Tensor(p.contents)

    def infotext_get(self, idx=0):
        "Get text describing the generation parameters."
        info = self._lib.mlis_infotext_get(self._ctx, idx)
        if info is None:
            raise RuntimeError("Failed to get infotext %d" % idx)
        return info.decode("utf8")

    def errstr_get(self):
        "Return an string describing the last error."
        e = self._lib.mlis_errstr_get(self._ctx)
        return e.decode("utf8") if e is not None else None

    # ---- direct use of the sub-models
    def clip_text_encode(self, text, features=False, no_norm=True, model_idx=4):
        t_embed = self._lib.mlis_tensor_get(self._ctx, MLIS_TENSOR_TMP)               # noqa: F821
        t_feat = self._lib.mlis_tensor_get(self._ctx, MLIS_TENSOR_TMP + 1) if features else None   # noqa: F821
        r = self._lib.mlis_clip_text_encode(self._ctx, text.encode("utf8"), t_embed, t_feat, model_idx, MLIS_CTEF_NO_NORM if no_norm else 0)
        if r < 0:
            raise RuntimeError("Failed to encode text with CLIP: %s" % (self.errstr_get()))
        embed = MLIS_Tensor(t_embed.contents)
        return (embed, MLIS_Tensor(t_feat.contents)) if features else embed

    def text_tokenize(self, text, model_idx=4):
        ptr = ctypes.POINTER(ctypes.c_int32)()
        n = self._lib.mlis_text_tokenize(self._ctx, text.encode("utf8"), ctypes.byref(ptr), model_idx)
        if n < 0:
            raise RuntimeError("Failed to tokenize: %s" % self.errstr_get())
        return [int(ptr[i]) for i in range(n)]

    def _transform(self, fn, what, src, flags):
        tin = src._c()
        tout = self._lib.mlis_tensor_get(self._ctx, MLIS_TENSOR_TMP + 2)              # noqa: F821
        if fn(self._ctx, ctypes.byref(tin), tout, flags) < 0:
            raise RuntimeError("Failed to %s: %s" % (what, self.errstr_get()))
        return MLIS_Tensor(tout.contents)

    def image_encode(self, image, flags=0):
        "VAE-encode an image tensor [1,3,H,W] in [0,1] into a latent."
        return self._transform(self._lib.mlis_image_encode, "encode the image", image, flags)

    def image_decode(self, latent, flags=0):
        "VAE-decode a latent into an image tensor in [0,1]."
        return self._transform(self._lib.mlis_image_decode, "decode the latent", latent, flags)

    def mask_encode(self, mask, flags=0):
        "Reduce a pixel mask to latent resolution."
        return self._transform(self._lib.mlis_mask_encode, "encode the mask", mask, flags)


def tensor_from_numpy(a):
    """float32 array of up to 4 dimensions -> MLIS_Tensor (n = shape reversed, padded with ones)."""
    import numpy as np
    a = np.ascontiguousarray(a, np.float32)
    t = MLIS_Tensor.__new__(MLIS_Tensor)
    t.n = tuple((list(a.shape[::-1]) + [1, 1, 1, 1])[:4])
    t.data = a.tobytes()
    return t


__all__ = [n for n in globals() if n.startswith("MLIS_") or n in ("MLImgSynth", "load_library", "tensor_from_numpy")]
