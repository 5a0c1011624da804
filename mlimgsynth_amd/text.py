"""Text conditioning for the generation driver: resident CLIP encoders and the SD1.5 / SDXL
cond + label assembly of mlis_text_cond_encode (reference src/mlimgsynth.c:1501-1563)."""
import ctypes

import numpy as np

from ._lib import check1, vp
from .engine import ClipParams, FP, MLCtx, _proto2, c_int, fptr


class ClipEncoderS(ctypes.Structure):
    _fields_ = [("C", vp), ("P", ClipParams), ("t_tokens", vp), ("t_embed", vp), ("n_prompt", ctypes.c_uint),
                ("want_feat", c_int), ("prefix", ctypes.c_char * 16), ("text_proj_host", vp)]


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
    """mlis_text_cond_encode (src/mlimgsynth.c:1501-1563) on resident encoders.

    SD1.x : cond = CLIP-L embed (clip_skip 1, final norm)                         [77][768]
    SDXL  : cond = CLIP-L embed (clip_skip 2, no norm) || CLIP-bigG embed (same)   [77][2048]
            label = bigG pooled feature (all layers + norm + text_proj) || size embeddings   [2816]
            empty negative prompt => uncond zeroed (uncond_empty_zero, mlimgsynth.c:1702-1703); unlabel is still
            computed from the empty prompt.
    tiny / tinyxl use the "tiny" tower so tests stay fast."""

    def __init__(self, model, width, height, seed=1234, stream=None):
        self.model, self.w, self.h = model, width, height
        if model in ("sd1", "tiny"):
            tower = "vit_l" if model == "sd1" else "tiny"
            self.enc = [ClipEncoder(tower, "clip", 1, clip_skip=1, norm=True, seed=seed, stream=stream)]
        elif model in ("sdxl", "tinyxl"):
            t1, t2 = ("vit_l", "vit_bigg") if model == "sdxl" else ("tiny", "tiny")
            self.enc = [ClipEncoder(t1, "clip", 1, clip_skip=2, norm=False, seed=seed, stream=stream),
                        ClipEncoder(t2, "clip2", 1, clip_skip=2, norm=False, seed=seed, stream=stream),
                        ClipEncoder(t2, "clip2", 1, want_feat=True, seed=seed, stream=stream)]
        else:
            raise ValueError(model)

    def encode(self, toks):
        """-> (cond [77][n_ctx], label [adm] or None)"""
        toks = np.asarray(toks, np.int32).reshape(1, -1)
        if len(self.enc) == 1:
            return self.enc[0].run(toks)[0][0], None
        e1 = self.enc[0].run(toks)[0][0]
        e2 = self.enc[1].run(toks)[0][0]
        feat = self.enc[2].run(toks)[1][0]
        cond = np.concatenate([e1, e2], axis=1)                 # mlimgsynth.c:1530-1539
        if self.model == "tinyxl":                              # tinyxl: n_ctx=128=64+64, adm=96=64+32 (shrunken size embedding)
            return cond, np.concatenate([feat, np.zeros(32, np.float32)])
        return cond, sdxl_label(feat, self.w, self.h)

    def flops(self):
        return sum(e.flops() for e in self.enc)

    def encode_pair(self, toks, neg_toks=()):
        """cond/label for the prompt and the (usually empty) negative prompt, with the SDXL zeroing rule."""
        cond, label = self.encode(toks)
        ncond, nlabel = self.encode(np.asarray(neg_toks, np.int32))
        if self.model in ("sdxl", "tinyxl") and len(neg_toks) == 0:
            ncond = np.zeros_like(ncond)
        return cond, label, ncond, nlabel


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
