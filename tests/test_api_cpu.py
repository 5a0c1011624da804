"""The public libmlimgsynth API (include/mlis_abi.h), parts that need no GPU: every prototype of the reference's FFI resolves,
the version gate, option parsing (mlis_option_set_str grammar incl. the "euler_a" shortcut), enum<->string helpers, error
reporting (errstr + handler), MLIS_Tensor helpers, mask down-sizing, model set-up in the dry runtime."""
import ctypes as C
import os

import numpy as np
import pytest

import mlis_ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mlimgsynth_amd import _lib
    _lib.lib()
    return F.bind(_lib.LIB_PATH)


def test_prototype_table_resolves_and_version_gate(lib):
    assert len(F.PROTOTYPES) >= 37
    assert not lib.mlis_ctx_create_i(0x000300) and not lib.mlis_ctx_create_i(0x000500)      # mlimgsynth.c:448
    m = F.Mlis(lib)
    m.close()
    assert not m.ctx.value                                                                   # destroy NULLs the handle


def test_enum_string_helpers(lib):
    assert lib.mlis_method_fromz(b"DPMPP2M") == 4 and lib.mlis_method_str(5) == b"dpmpp2s" and lib.mlis_method_fromz(b"nope") == -1
    assert lib.mlis_sched_fromz(b"karras") == 2 and lib.mlis_stage_desc(4) == b"Denoising" and lib.mlis_stage_str(9) == b"???"
    assert lib.mlis_option_fromz(b"cfg-scale") == lib.mlis_option_fromz(b"CFG_SCALE") == 12 and lib.mlis_option_str(24) == b"seed"
    assert lib.mlis_model_type_fromz(b"sdxl") == 3 and lib.mlis_model_type_desc(1) == b"Stable Diffusion 1.x"
    assert lib.mlis_loglvl_fromz(b"debug") == 50 and lib.mlis_loglvl_str(30) == b"info"


def test_option_grammar_and_errors(lib):
    m = F.Mlis(lib)
    seen = []
    H = F.ERRHANDLER(lambda ud, ctx, ei: seen.append((ei.contents.code, ei.contents.desc.decode())))
    assert lib.mlis_option_set(m.ctx, F.OPT["ERROR_HANDLER"], H, None) == 1
    m.set("image-dim", 512, 768)
    m.set("cfg-scale", 7.5)
    m.set("method", "euler_a")                      # shortcut: euler + s_ancestral 1 (mlimgsynth_options_set.c.h:88-99)
    m.set("SCHEDULER", "karras")
    m.set("steps", 12)
    m.set("no_decode", "yes")
    m.set("seed", 42)
    m.set("prompt", "a (dog:1.5), with commas, kept whole")
    assert lib.mlis_option_set(m.ctx, F.OPT["CFG_SCALE"], C.c_double(3.0)) == 1
    assert lib.mlis_option_set(m.ctx, F.OPT["IMAGE_DIM"], 64, 64) == 1
    p = C.c_char_p()
    assert lib.mlis_option_get(m.ctx, F.OPT["PROMPT"], C.byref(p)) == 1 and p.value == b"a (dog:1.5), with commas, kept whole"
    for name, val, code in [("steps", "12x", -4), ("cfg_scale", "300", -4), ("method", "rk4", -4), ("method", "rk4_a", -4),
                            ("nosuchoption", "1", -3), ("no_decode", "maybe", -4), ("prompt", "unbalanced )", -5), ("image", "x", -4),
                            ("weight_type", "q8_0", -4)]:
        r = lib.mlis_option_set_str(m.ctx, name.encode(), val.encode())
        assert r == code, (name, val, r, m.err())
    assert len(seen) == 9 and seen[0][0] == -4 and "12x" in seen[0][1]
    assert lib.mlis_option_set(m.ctx, 99, 1) == -3
    with pytest.raises(RuntimeError):
        m.set("lora", "some_lora", 0.8)             # accepted option, reported as not implemented
    m.close()


def test_tensor_helpers_and_mask_encode(lib):
    a, b = F.Tensor(), F.Tensor()
    lib.mlis_tensor_resize(C.byref(a), 16, 8, 1, 1)
    assert lib.mlis_tensor_count(C.byref(a)) == 128
    arr = np.ctypeslib.as_array(a.d, shape=(8, 16))
    arr[:] = np.arange(128, dtype=np.float32).reshape(8, 16)
    lib.mlis_tensor_copy(C.byref(b), C.byref(a))
    assert abs(lib.mlis_tensor_similarity(C.byref(a), C.byref(b)) - 1.0) < 1e-6
    m = F.Mlis(lib)
    lm = F.Tensor()
    assert lib.mlis_mask_encode(m.ctx, C.byref(a), C.byref(lm), 0) == 1
    got = F.tensor_np(lm)[0, 0]
    ref = arr.reshape(1, 8, 2, 8).mean(axis=(1, 3))                   # ltensor_downsize: 8x8 box average
    assert got.shape == (1, 2) and np.allclose(got, ref)
    for t in (a, b, lm):
        lib.mlis_tensor_free(C.byref(t))
    m.close()


def test_setup_synthetic_and_missing_models_in_dry_runtime(lib):
    lib.mlsd_runtime_dry(1)
    try:
        m = F.Mlis(lib)
        assert lib.mlis_setup(m.ctx) == -6 and "no model set" in m.err()
        m.set("model", "/nonexistent/model.safetensors")
        assert lib.mlis_setup(m.ctx) == -6
        m.set("model", "synth:sdxl")
        assert lib.mlis_setup(m.ctx) == 1
        mt = C.c_int()
        assert lib.mlis_option_get(m.ctx, F.OPT["MODEL_TYPE"], C.byref(mt)) == 1 and mt.value == 3
        m.set("backend", "CUDA0")
        assert lib.mlis_setup(m.ctx) < 0 and "not available" in m.err()
        m.close()
    finally:
        lib.mlsd_runtime_dry(0)


def test_python_wrapper_constants_match_the_header_and_errors_raise():
    """mlimgsynth_amd/mlimgsynth.py (counterpart of python/mlimgsynth.py): every MLIS_* constant it defines equals the enumerator of
    include/mlis_abi.h with that name (MLIS_MODEL_x = MLIS_SUBMODEL_x in the header), options work through both entry points,
    failures raise RuntimeError carrying the library's error text."""
    import re
    from mlimgsynth_amd import mlimgsynth as W
    hdr = open(os.path.join(ROOT, "include", "mlis_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    enums = {}
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}", hdr, flags=re.S):
        val = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [x.strip() for x in item.split("=", 1)]
                val = eval(expr, {}, dict(enums))
            else:
                name, val = item, val + 1
            enums[name] = val
    checked = 0
    for name in dir(W):
        if not name.startswith("MLIS_") or not isinstance(getattr(W, name), int) or name in ("MLIS_VERSION",):
            continue
        hname = name.replace("MLIS_MODEL_", "MLIS_SUBMODEL_") if re.match(r"MLIS_MODEL_(NONE|UNET|VAE|TAE|CLIP|CLIP2)$", name) else name
        assert hname in enums, name
        assert enums[hname] == getattr(W, name), name
        checked += 1
    assert checked >= 90 and W.MLIS_OPT_NO_PROMPT_PARSE == 35 and W.MLIS_TENSOR_TMP == 0x100
    with W.MLImgSynth() as m:
        m.option_set(W.MLIS_OPT_IMAGE_DIM, 512, 768)        # by id (variadic)
        m.option_set("cfg-scale", 7.0)                      # by name (string form)
        m.option_set(W.MLIS_OPT_CFG_SCALE, 3.5)
        m.option_set(W.MLIS_OPT_PROMPT, "a cat")
        v = C.c_char_p()
        m.option_get(W.MLIS_OPT_PROMPT, v)                  # the reference implements option_get for 4 options (mlimgsynth_options_get.c.h)
        assert v.value == b"a cat"
        with pytest.raises(RuntimeError, match="unknown option"):
            m.option_get(W.MLIS_OPT_CFG_SCALE, C.c_double())
        with pytest.raises(RuntimeError, match="steps"):
            m.option_set("steps", "12x")
        with pytest.raises(RuntimeError):
            m.option_set(3.5)
        with pytest.raises(RuntimeError, match="image"):
            m.image_get(0)
        a = W.tensor_from_numpy(np.array([1.0, 0.0, 2.0], np.float32))
        b = W.tensor_from_numpy(np.array([2.0, 0.0, 4.0], np.float32))
        assert abs(a.similarity(b) - 1.0) < 1e-6 and a.n == (3, 1, 1, 1)
