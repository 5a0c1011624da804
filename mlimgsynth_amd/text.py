"""Text conditioning for the generation driver: resident CLIP encoders and the SD1.5 / SDXL
cond + label assembly of mlis_text_cond_encode (reference src/mlimgsynth.c:1501-1563)."""
import ctypes

import numpy as np

from ._lib import check1, vp
from .engine import ClipParams, FP, MLCtx, _proto2, c_int, fptr


class ClipEncoderS(ctypes.Structure):
    _fields_ = [("C", vp), ("P", ClipParams), ("t_tokens", vp), ("t_embed", vp), ("n_prompt", ctypes.c_uint),
                ("want_feat", c_int), ("prefix", ctypes.c_char * 16), ("text_proj_host", vp), ("t_tap", vp), ("tap_dev", vp)]


def _l():
    l = _proto2()
    if not getattr(l, "_text_done", False):
        l.clip_encoder_init.argtypes = [ctypes.POINTER(ClipEncoderS), vp, ctypes.POINTER(ClipParams), ctypes.c_char_p,
                                        ctypes.c_uint, c_int, ctypes.c_bool, ctypes.c_bool]
        l.clip_encoder_run.argtypes = [ctypes.POINTER(ClipEncoderS), ctypes.c_uint, ctypes.POINTER(ctypes.c_int32), FP, FP]
        l.clip_encoder_free.argtypes = [ctypes.POINTER(ClipEncoderS)]
        l.sdxl_label_build.argtypes = [FP, c_int, c_int, c_int, FP, c_int]
        l._text_done = True
    return l


class ClipEncoder:
    """Resident CLIP text tower: graph and weights stay in HBM between prompts."""

    def __init__(self, model, prefix, n_prompt=1, clip_skip=1, norm=True, want_feat=False, seed=1234, stream=None):
        l = _l()
        self.P = ClipParams()
        check1(l.clip_params_get(model.encode(), ctypes.byref(self.P)), "clip_params_get")
        self.ctx = MLCtx(stream)
        self.E = ClipEncoderS()
        self.n_prompt, self.want_feat = n_prompt, want_feat
        check1(l.clip_encoder_init(ctypes.byref(self.E), self.ctx.h, ctypes.byref(self.P), prefix.encode(), n_prompt,
                                   clip_skip, norm, want_feat), "clip_encoder_init")
        self.ctx.params_synth(seed)

    def run(self, toks):
        toks = np.ascontiguousarray(toks, np.int32).reshape(self.n_prompt, -1)
        embed = np.empty((self.n_prompt, self.P.n_token, self.P.d_embed), np.float32)
        feat = np.empty((self.n_prompt, self.P.d_embed), np.float32) if self.want_feat else None
        check1(_l().clip_encoder_run(ctypes.byref(self.E), toks.shape[1], toks.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                     fptr(embed), fptr(feat)), "clip_encoder_run")
        return embed, feat

    def flops(self):
        return self.ctx.info().flops

    def destroy(self):
        if self.ctx is not None:
            _l().clip_encoder_free(ctypes.byref(self.E))
            self.ctx.destroy()
            self.ctx = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def sdxl_label(feat, width, height):
    feat = np.ascontiguousarray(feat, np.float32).reshape(-1)
    out = np.empty(feat.size + 1536, np.float32)
    check1(_l().sdxl_label_build(fptr(feat), feat.size, width, height, fptr(out), out.size), "sdxl_label_build")
    return out


class TextConditioner:
    """mlis_text_cond_encode (src/mlimgsynth.c:1501-1563) on resident encoders: thin wrapper over the C object
    MLIS_AmdTextCond (csrc/host/textcond.c), which holds the towers and does the SD1.5 / SDXL assembly."""

    def __init__(self, model, width, height, seed=1234, stream=None):
        l = _proto2()
        I32P = ctypes.POINTER(ctypes.c_int32)
        l.mlis_amd_textcond_create.restype = vp
        l.mlis_amd_textcond_create.argtypes = [ctypes.c_char_p, c_int, c_int, ctypes.c_uint64, vp]
        l.mlis_amd_textcond_destroy.argtypes = [vp]
        l.mlis_amd_textcond_dims.argtypes = [vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
        l.mlis_amd_textcond_flops.argtypes = [vp]
        l.mlis_amd_textcond_flops.restype = ctypes.c_double
        l.mlis_amd_textcond_encode.argtypes = [vp, I32P, c_int, FP, FP]
        l.mlis_amd_textcond_encode_pair.argtypes = [vp, I32P, c_int, I32P, c_int, FP, FP, FP, FP]
        self._l, self.model = l, model
        h = l.mlis_amd_textcond_create(model.encode(), width, height, seed, vp(stream))
        if not h:
            check1(-1, "mlis_amd_textcond_create")
        self.h = vp(h)
        a, b = c_int(), c_int()
        l.mlis_amd_textcond_dims(self.h, ctypes.byref(a), ctypes.byref(b))
        self.n_ctx, self.n_label = a.value, b.value

    @staticmethod
    def _toks(t):
        t = np.ascontiguousarray(np.asarray(t, np.int32).reshape(-1))
        return t, t.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if t.size else None

    def encode(self, toks):
        """-> (cond [77][n_ctx], label [n_label] or None)"""
        t, tp = self._toks(toks)
        cond = np.empty((77, self.n_ctx), np.float32)
        label = np.empty(self.n_label, np.float32) if self.n_label else None
        check1(self._l.mlis_amd_textcond_encode(self.h, tp, t.size, fptr(cond), fptr(label)), "mlis_amd_textcond_encode")
        return cond, label

    def encode_pair(self, toks, neg_toks=()):
        """cond/label for the prompt and the (usually empty) negative prompt, with the SDXL zeroing rule."""
        t, tp = self._toks(toks)
        n, np_ = self._toks(neg_toks)
        cond, ncond = np.empty((77, self.n_ctx), np.float32), np.empty((77, self.n_ctx), np.float32)
        label = np.empty(self.n_label, np.float32) if self.n_label else None
        nlabel = np.empty(self.n_label, np.float32) if self.n_label else None
        check1(self._l.mlis_amd_textcond_encode_pair(self.h, tp, t.size, np_, n.size, fptr(cond), fptr(label), fptr(ncond), fptr(nlabel)),
               "mlis_amd_textcond_encode_pair")
        return cond, label, ncond, nlabel

    def flops(self):
        return self._l.mlis_amd_textcond_flops(self.h)

    def destroy(self):
        if self.h:
            self._l.mlis_amd_textcond_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ClipTokenizer:
    """CLIP BPE tokenizer (csrc/host/clip_tokenizer.c; reference clip_tokenize, src/clip.c:59-278).

    The merge table is data the user supplies: `from_pairs` (int32 [n][2], rank order) or `from_file` (OpenAI's
    bpe_simple_vocab_16e6.txt / a merges.txt)."""

    def __init__(self):
        l = _proto2()
        l.clip_tokr_new.restype = vp
        l.clip_tokr_free.argtypes = [vp]
        l.clip_tokr_set_merges.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), c_int]
        l.clip_tokr_load_merges_txt.argtypes = [vp, ctypes.c_char_p, c_int]
        l.clip_tokr_n_vocab.argtypes = [vp]
        l.clip_tokr_n_merges.argtypes = [vp]
        l.clip_tokenize.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), c_int]
        l.clip_token_decode.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p, c_int, ctypes.POINTER(c_int)]
        self._l = l
        self.h = vp(l.clip_tokr_new())

    @classmethod
    def from_pairs(cls, pairs):
        t = cls()
        p = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        check1(t._l.clip_tokr_set_merges(t.h, p.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(p)), "clip_tokr_set_merges")
        return t

    @classmethod
    def from_file(cls, path, max_merges=0):
        t = cls()
        r = t._l.clip_tokr_load_merges_txt(t.h, str(path).encode(), max_merges)
        if r < 0:
            check1(r, "clip_tokr_load_merges_txt")
        return t

    @property
    def n_vocab(self):
        return self._l.clip_tokr_n_vocab(self.h)

    @property
    def n_merges(self):
        return self._l.clip_tokr_n_merges(self.h)

    def encode(self, text):
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        out = np.empty(max(len(raw), 1), np.int32)       # a word never yields more tokens than bytes
        n = self._l.clip_tokenize(self.h, raw, len(raw), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), out.size)
        if n < 0:
            check1(n, "clip_tokenize")
        return [int(v) for v in out[:n]]

    def token_bytes(self, token):
        buf = ctypes.create_string_buffer(256)
        eow = c_int(0)
        n = self._l.clip_token_decode(self.h, token, buf, 256, ctypes.byref(eow))
        if n < 0:
            check1(n, "clip_token_decode")
        return buf.raw[:n], bool(eow.value)

    def decode(self, tokens):
        out = bytearray()
        for t in tokens:
            b, eow = self.token_bytes(t)
            out += b
            if eow:
                out += b" "
        return out.decode("utf-8", errors="replace").rstrip()

    def __del__(self):
        try:
            if self.h:
                self._l.clip_tokr_free(self.h)
                self.h = None
        except Exception:
            pass
