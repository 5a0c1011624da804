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


# ---------------------------------------------------------------------------------------------------------------- GGUF
def _gguf_api(L):
    L.mlts_open.restype = ctypes.c_void_p
    L.mlts_open.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.mlts_entry_to_f32.argtypes = [ctypes.POINTER(TSEntry), ctypes.POINTER(ctypes.c_float), ctypes.c_int64]
    return L


def _entry_f32(L, e):
    n = int(np.prod([e.contents.shape[i] for i in range(4)]))
    out = np.empty(n, np.float32)
    assert L.mlts_entry_to_f32(e, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n) == 1
    return out


@pytest.mark.parametrize("version", [2, 3])
def test_gguf_index_metadata_and_every_tensor_type(L, tmp_path, version):
    """GGUF v2/v3 container (src/ccompute/tensorstore_gguf.c:171-235): metadata of every value type is skipped, tensor
    dims are fastest-first, data starts at the next multiple of 32 after the tensor table; F32/F16/BF16 and the block types
    Q8_0/Q4_0/Q4_1/Q5_0/Q5_1 convert to fp32 by the published block formulas (expected values computed by the test's own
    quantiser from the integer codes it wrote)."""
    import gguf_io as G
    _gguf_api(L)
    rng = np.random.default_rng(version)
    tensors = [("a.f32", rng.standard_normal((3, 5)).astype(np.float32), "F32"),
               ("b.f16", rng.standard_normal((2, 3, 3, 4)).astype(np.float32), "F16"),
               ("c.bf16", rng.standard_normal((7,)).astype(np.float32) * 100, "BF16")]
    for kind in ("Q8_0", "Q4_0", "Q4_1", "Q5_0", "Q5_1"):
        tensors.append((f"w.{kind.lower()}", rng.standard_normal((6, 64)).astype(np.float32) * rng.uniform(0.1, 3), kind))
    tensors.append(("z.zero", np.zeros((2, 32), np.float32), "Q4_0"))
    meta = [("general.architecture", G.T_STR, "sd"), ("general.alignment", G.T_U32, 32), ("x.u8", G.T_U8, 7), ("x.i8", G.T_I8, -3),
            ("x.u16", G.T_U16, 65000), ("x.i16", G.T_I16, -300), ("x.i32", G.T_I32, -70000), ("x.f32", G.T_F32, 1.5), ("x.bool", G.T_BOOL, 1),
            ("x.u64", G.T_U64, 2 ** 40), ("x.i64", G.T_I64, -2 ** 40), ("x.f64", G.T_F64, 2.25),
            ("x.arr_i32", G.T_ARR, (G.T_I32, [1, 2, 3])), ("x.arr_str", G.T_ARR, (G.T_STR, ["ab", "", "cde"])), ("x.arr_empty", G.T_ARR, (G.T_F32, []))]
    path = str(tmp_path / f"t{version}.gguf")
    expect = G.write(path, tensors, meta, version)
    S = L.mlts_open(path.encode(), 0)
    assert S, L.mlsd_last_error()
    assert L.mlts_count(S) == len(tensors)
    for name, arr, kind in tensors:
        e = L.mlts_find(S, name.encode())
        assert e and e.contents.dtype == G.GGML[kind] and e.contents.n_dim == arr.ndim
        assert [e.contents.shape[i] for i in range(4)] == (list(arr.shape[::-1]) + [1, 1, 1, 1])[:4]
        got = _entry_f32(L, e).reshape(arr.shape)
        assert np.array_equal(got, expect[name]), name                                   # bit-exact: fp16 scale x small integer (+ fp16 min) in fp32
        if kind.startswith("Q"):
            step = np.abs(arr).max() / {"Q8_0": 127, "Q4_0": 7, "Q4_1": 7.5, "Q5_0": 15, "Q5_1": 15.5}[kind]
            assert np.abs(got - arr).max() <= 1.01 * step + 1e-3 * np.abs(arr).max(), kind   # and the codes do describe the data
    L.mlts_close(S)


def test_gguf_checkpoint_loads_like_safetensors(L, tmp_path):
    """the same tiny checkpoint as safetensors and as GGUF (checkpoint-side names, fused open_clip in_proj, F16 + Q8_0 linear
    weights): same index after name conversion + QKV split; the plan loads from it in the dry runtime"""
    import gguf_io as G
    from safetensors.numpy import load_file
    from mlimgsynth_amd import _lib, engine
    _gguf_api(L)
    st = str(tmp_path / "m.safetensors")
    LC.write_checkpoint(st, "tinyxl", "F16")
    src = load_file(st)
    tensors = []
    for k, v in src.items():
        kind = "F32" if v.dtype == np.float32 else "F16"
        if kind == "F16" and v.ndim == 2 and v.shape[1] % 32 == 0 and "attn" in k and "in_proj" not in k:
            kind = "Q8_0"
        tensors.append((k, v.astype(np.float32), kind))
    gg = str(tmp_path / "m.gguf")
    expect = G.write(gg, tensors, [("general.architecture", G.T_STR, "sdxl")])
    A, B = L.mlts_open(st.encode(), 1), L.mlts_open(gg.encode(), 1)
    assert A and B, L.mlsd_last_error()
    assert L.mlts_count(A) == L.mlts_count(B)
    nq = 0
    for i in range(L.mlts_count(A)):
        ea = L.mlts_at(A, i)
        eb = L.mlts_find(B, ea.contents.name)
        assert eb and [ea.contents.shape[j] for j in range(4)] == [eb.contents.shape[j] for j in range(4)]
        fa, fb = _entry_f32(L, ea), _entry_f32(L, eb)
        if eb.contents.dtype == G.GGML["Q8_0"]:
            nq += 1
            assert np.abs(fa - fb).max() <= np.abs(fa).max() / 127 * 0.51 + 1e-6
        else:
            assert np.array_equal(fa, fb), ea.contents.name
    assert nq > 10
    L.mlsd_runtime_dry(1)
    try:
        un = engine.Unet("tinyxl", 8, 8, 2, synth=False)
        assert L.mlctx_tstore_load(un.ctx.h, B) == len(LC.model_params("tinyxl")) - sum(1 for k, _, _ in LC.model_params("tinyxl") if not k.startswith("unet."))
        un.ctx.destroy()
    finally:
        L.mlsd_runtime_dry(0)
        L.mlts_close(A); L.mlts_close(B)


def test_gguf_malformed_files_fail_loudly(L, tmp_path):
    import struct
    import gguf_io as G
    _gguf_api(L)
    good = str(tmp_path / "g.gguf")
    G.write(good, [("a", np.ones((2, 32), np.float32), "Q8_0")], [("k", G.T_U32, 1)])
    raw = open(good, "rb").read()

    def opens(b, name):
        p = str(tmp_path / name)
        open(p, "wb").write(b)
        return L.mlts_open(p.encode(), 0)
    assert opens(raw, "ok.gguf")
    assert not opens(raw[:4] + struct.pack("<I", 1) + raw[8:], "v1.gguf") and b"version" in L.mlsd_last_error()      # tensorstore_gguf.c:190-191
    assert not opens(raw[:40], "trunc.gguf")
    assert not opens(raw[:-40], "short_data.gguf") and b"outside the file" in L.mlsd_last_error()
    bad_type = raw.replace(struct.pack("<IQ", 8, 0), struct.pack("<IQ", 99, 0))
    assert not opens(bad_type, "type.gguf") and b"unknown tensor type" in L.mlsd_last_error()
    # 33 elements of a block type: not whole blocks
    p = str(tmp_path / "blk.gguf")
    head = b"GGUF" + struct.pack("<IQQ", 3, 1, 0) + struct.pack("<Q", 1) + b"a" + struct.pack("<IQIQ", 1, 33, 8, 0)
    open(p, "wb").write(head + b"\0" * 128)
    assert not L.mlts_open(p.encode(), 0) and b"whole blocks" in L.mlsd_last_error()
