"""Checkpoint loading, CPU part: tensor-name conversion against vectors produced by the
REFERENCE's own tensor_name_conv.c (tests/golden/name_conv.json.gz, and live against oracle/_ref when it is built), the
safetensors index, the open_clip in_proj split, model identification and dtype conversion.  The upload itself runs in
the dry runtime (device memory served from host memory)."""
import ctypes
import gzip
import json
import os

import numpy as np
import pytest

import loader_cases as LC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TSEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("dtype", ctypes.c_int), ("n_dim", ctypes.c_int), ("shape", ctypes.c_int64 * 4),
                ("size", ctypes.c_size_t), ("data", ctypes.c_void_p)]


@pytest.fixture(scope="module")
def L():
    from mlimgsynth_amd import _lib
    l = _lib.lib()
    l.tnconv_sd.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    l.mlts_open_safetensors.restype = ctypes.c_void_p
    l.mlts_open_safetensors.argtypes = [ctypes.c_char_p, ctypes.c_int]
    l.mlts_close.argtypes = [ctypes.c_void_p]
    l.mlts_count.argtypes = [ctypes.c_void_p]
    l.mlts_find.restype = ctypes.POINTER(TSEntry)
    l.mlts_find.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    l.mlts_at.restype = ctypes.POINTER(TSEntry)
    l.mlts_at.argtypes = [ctypes.c_void_p, ctypes.c_int]
    l.mlts_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    l.mlts_model_identify.restype = ctypes.c_char_p
    l.mlts_model_identify.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    l.mlctx_tstore_load.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return l


def conv(L, name):
    buf = ctypes.create_string_buffer(1024)
    r = L.tnconv_sd(name.encode(), buf, 1024)
    return [r, buf.value.decode()]


def test_name_conversion_matches_reference_vectors(L):
    """4431 names: every tensor of the SD1.5 / SD2 / SDXL single-file layouts, separator variants, diffusers-style blocks,
    tensors that must be dropped; expected values were produced by the reference's own tnconv_sd"""
    with gzip.open(os.path.join(ROOT, "tests", "golden", "name_conv.json.gz"), "rt") as f:
        vec = json.load(f)
    assert len(vec) > 4000
    bad = [(n, conv(L, n), v) for n, v in vec.items() if conv(L, n) != v]
    assert not bad, bad[:5]


def test_name_conversion_live_against_reference_build(L):
    so = os.path.join(ROOT, "oracle", "_ref", "libtnconv_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtnconv_ref.so not built (the reference is only present in the build container)")
    ref = ctypes.CDLL(so)
    buf = ctypes.create_string_buffer(1024)
    import ckpt_names as CN
    n = 0
    for model in ("tiny", "tinyxl"):
        for k, _, _ in LC.model_params(model):
            e = CN.external_name(k, "sd1" if model == "tiny" else "sdxl")
            e = e[1] if isinstance(e, tuple) else e
            r = ref.ref_tnconv_sd(e.encode(), buf, 1024)
            assert conv(L, e) == [r, buf.value.decode()], e
            if r == 1:
                assert buf.value.decode() == k          # and the generator's inverse mapping round-trips
            n += 1
    assert n > 500


@pytest.fixture(scope="module")
def tinyxl_ckpt(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("ckpt") / "tinyxl_f16.safetensors")
    n = LC.write_checkpoint(p, "tinyxl", "F16")
    return p, n


def test_safetensors_index_and_qkv_split(L, tinyxl_ckpt):
    path, n_file = tinyxl_ckpt
    S = L.mlts_open_safetensors(path.encode(), 1)
    assert S
    params = LC.model_params("tinyxl")
    un, sp = ctypes.c_int(), ctypes.c_int()
    L.mlts_stats(S, ctypes.byref(un), ctypes.byref(sp))
    assert un.value == 1                                   # model_ema.decay dropped
    assert sp.value == 2 * 3                               # clip2: 3 layers x (in_proj_weight, in_proj_bias)
    assert L.mlts_count(S) == len(params)
    for k, f16, shape in params:
        e = L.mlts_find(S, k.encode())
        assert e, k
        e = e.contents
        assert e.dtype == (1 if f16 else 0), k
        cnt = int(np.prod(shape))
        assert e.shape[0] * e.shape[1] * e.shape[2] * e.shape[3] == cnt, k
        raw = (ctypes.c_char * e.size).from_address(e.data)
        got = np.frombuffer(raw, np.float16 if f16 else np.float32).astype(np.float32)
        assert np.array_equal(got, LC.synth_values(k, shape, f16).reshape(-1)), k     # incl. the q/k/v thirds of in_proj
    wt = ctypes.c_int(-1)
    assert L.mlts_model_identify(S, ctypes.byref(wt)) is None     # tinyxl's n_ctx 128 is not a known model: reported, not guessed
    L.mlts_close(S)


def test_tstore_load_in_dry_runtime(L, tinyxl_ckpt):
    from mlimgsynth_amd import _lib, engine
    path, _ = tinyxl_ckpt
    L.mlsd_runtime_dry(1)
    try:
        S = L.mlts_open_safetensors(path.encode(), 1)
        un = engine.Unet("tinyxl", 8, 8, 1, synth=False)
        n = L.mlctx_tstore_load(un.ctx.h, S)
        assert n == len(un.ctx.param_list())
        un2 = engine.Unet("tiny", 8, 8, 1, synth=False)                      # a different model: a tensor is missing or mis-sized
        assert L.mlctx_tstore_load(un2.ctx.h, S) < 0
        assert "not found" in _lib.last_error() or "elements" in _lib.last_error()
        L.mlts_close(S)
    finally:
        L.mlsd_runtime_dry(0)


def test_model_identification_probe_tensor(L, tmp_path):
    """mlis_model_identify keys on one cross-attention k_proj tensor (src/mlimgsynth.c:1206-1249)"""
    from safetensors.numpy import save_file
    cases = [("model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight", (320, 768), np.float16, b"sd1", 1),
             ("model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight", (320, 1024), np.float32, b"sd2", 0),
             ("model.diffusion_model.input_blocks.4.1.transformer_blocks.0.attn2.to_k.weight", (640, 2048), np.float16, b"sdxl", 1)]
    for i, (name, shape, dt, want, wtype) in enumerate(cases):
        p = str(tmp_path / f"probe{i}.safetensors")
        save_file({name: np.zeros(shape, dt)}, p)
        S = L.mlts_open_safetensors(p.encode(), 1)
        wt = ctypes.c_int(-1)
        assert L.mlts_model_identify(S, ctypes.byref(wt)) == want and wt.value == wtype
        L.mlts_close(S)


def test_malformed_files_fail_loudly(L, tmp_path):
    from mlimgsynth_amd import _lib
    p = tmp_path / "bad.safetensors"
    p.write_bytes(b"\x10\x00\x00\x00\x00\x00\x00\x00{\"a\":{\"dtype\":1}}")
    assert not L.mlts_open_safetensors(str(p).encode(), 1)
    assert "malformed" in _lib.last_error() or "invalid" in _lib.last_error()
    assert not L.mlts_open_safetensors(str(tmp_path / "missing.safetensors").encode(), 1)
    assert "could not open" in _lib.last_error()
