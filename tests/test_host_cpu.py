"""CPU-only checks of the product's host side (no compute calls without a GPU):
  * the C-ABI library loads and exports every symbol that include/*.h declares;
  * the plan builder, run in "dry" mode (device memory served from host memory, kernel launches
    refused), produces the reference's parameter names, parameter counts and the algorithmic FLOP
    totals of SURVEY.md App. C for the real SD1.5 / SDXL / VAE / TAE / CLIP configurations;
  * host arithmetic that needs no GPU (Philox, schedule, fp16 conversion) matches the oracle / KATs;
  * compute without a GPU fails loudly (no CPU fallback in the product path).
"""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def dry():
    from mlimgsynth_amd import _lib, engine
    L = _lib.lib()
    L.mlsd_runtime_dry(1)
    yield engine
    L.mlsd_runtime_dry(0)


def declared_symbols():
    syms = set()
    for h in ("mlsd_kernels.h", "mlblock_amd.h", "mlimgsynth_amd.h", "mlis_abi.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"static inline[^{]*\{[^}]*\}", "", src)
        src = re.sub(r"typedef[^;{]*\(\s*\*[^;]*;", "", src)          # function-pointer typedefs are not symbols
        src = re.sub(r"^\s*#\s*define[^\n]*$", "", src, flags=re.M)   # macros (mlis_ctx_create()) are not symbols
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src):
            name = m.group(1)
            if name not in ("defined", "sizeof"):
                syms.add(name)
    return syms


def test_cabi_exports_every_declared_symbol():
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) > 90
    missing = [s for s in sorted(syms) if not hasattr(L, s)]
    assert not missing, missing


def test_compute_without_gpu_fails_loudly(dry):
    from mlimgsynth_amd import _lib
    un = dry.Unet("tiny", 8, 8, 1, synth=False)
    with pytest.raises(_lib.MlsdError):
        un.ctx.params_synth(1234)            # would launch a kernel: refused, no silent CPU path
    with pytest.raises(_lib.MlsdError):
        un.ctx.compute()


@pytest.mark.parametrize("model,lat,flops_T,params_M", [("sd1", 64, 0.803, 859.5), ("sdxl", 128, 6.761, 2567.5)])
def test_unet_plan_matches_survey_counts(dry, model, lat, flops_T, params_M):
    un = dry.Unet(model, lat, lat, 1, synth=False)
    info = un.ctx.info()
    npar = sum(int(np.prod(ne)) for _, _, ne in un.ctx.param_list())
    assert abs(info.flops / 1e12 - flops_T) < 0.002            # SURVEY.md §8d / App. C (2*MAC of conv, linear, QK^T, PV)
    assert abs(npar / 1e6 - params_M) < 0.2
    keys = {k for k, _, _ in un.ctx.param_list()}
    for k in ["unet.in.conv.weight", "unet.time_embed.0.weight", "unet.in.1.0.emb_proj.weight", "unet.mid.1.transf.0.attn2.k_proj.weight",
              "unet.out.0.0.skip_conv.weight", "unet.out.norm.weight", "unet.out.conv.bias"]:
        assert k in keys, k
    if model == "sdxl":
        assert "unet.label_embed.0.weight" in keys and "unet.in.7.1.transf.9.ff.net.0.proj.weight" in keys
        assert "unet.in.1.1.norm.weight" not in keys             # SDXL has no attention at the first level


def test_unet_param_names_equal_oracle_names(dry):
    un = dry.Unet("tinyxl", 8, 8, 2, synth=False)
    U, P = O.unet_params("tinyxl"), O.Params(1)
    x = np.zeros((1, 4, 8, 8), np.float32)
    ctx = np.zeros((1, 1, 77, U.n_ctx), np.float32)
    lab = np.zeros((1, 1, 1, U.ch_adm_in), np.float32)
    O.from_ot(O.L().orc_unet_graph(P.h, b"unet", U, O.to_ot(x), 1.0, O.to_ot(ctx), O.to_ot(lab)))
    mine = {k: int(np.prod(ne)) for k, _, ne in un.ctx.param_list()}
    theirs = {k: int(np.prod(ne)) for k, _, ne in P.names()}
    assert mine == theirs


def test_decoder_and_clip_plans_match_survey_counts(dry):
    dec = dry.Decoder.__new__(dry.Decoder)            # build without loading weights (needs a GPU)
    l = dry._proto2()
    dec.ctx, dec.t_lat = dry.MLCtx(), dry.vp()
    vp_ = dry.VaeParams()
    l.vae_params_get(b"sdxl", ctypes.byref(vp_))
    assert abs(vp_.scale_factor - 0.13025) < 1e-7                  # src/vae.c:43
    assert l.sdvae_decode_init(dec.ctx.h, ctypes.byref(vp_), 128, 128, 1, ctypes.byref(dec.t_lat)) == 1
    assert l.sdvae_decode_build(dec.ctx.h, ctypes.byref(vp_), dec.t_lat) == 1
    info = dec.ctx.info()
    assert abs(info.flops / 1e12 - 10.470) < 0.01                  # VAE decode 1024^2: 10.470 TFLOP
    assert abs(sum(int(np.prod(ne)) for _, _, ne in dec.ctx.param_list()) / 1e6 - 49.5) < 0.1
    tae = dry.MLCtx()
    tl = dry.vp()
    assert l.sdtae_decode_init(tae.h, 128, 128, 1, ctypes.byref(tl)) == 1 and l.sdtae_decode_build(tae.h, tl) == 1
    assert abs(tae.info().flops / 1e12 - 0.565) < 0.002            # TAE decode 1024^2: 0.565 TFLOP
    keys = {k for k, _, _ in tae.param_list()}
    assert {"tae.decoder.layers.0.weight", "tae.decoder.layers.2.conv.0.weight", "tae.decoder.layers.6.weight",
            "tae.decoder.layers.17.conv.4.bias", "tae.decoder.layers.18.weight"} <= keys
    assert "tae.decoder.layers.6.bias" not in keys                 # the post-upsample convs have no bias (src/tae.c:83-84)
    from mlimgsynth_amd import text
    enc = text.ClipEncoder.__new__(text.ClipEncoder)
    enc.P, enc.ctx, enc.E = dry.ClipParams(), dry.MLCtx(), text.ClipEncoderS()
    text._l().clip_params_get(b"vit_l", ctypes.byref(enc.P))
    assert text._l().clip_encoder_init(ctypes.byref(enc.E), enc.ctx.h, ctypes.byref(enc.P), b"clip", 1, 1, True, False) == 1
    npar = sum(int(np.prod(ne)) for _, _, ne in enc.ctx.param_list())
    assert abs(npar / 1e6 - 123.0) < 0.2                           # CLIP-L text tower 123.0 M parameters
    assert abs(enc.ctx.info().flops / 1e12 - 0.013) < 0.001
    keys = {k for k, _, _ in enc.ctx.param_list()}
    assert {"clip.text.embed.token.weight", "clip.text.embed.position.weight", "clip.text.encoder.layers.11.mlp.fc2.bias",
            "clip.text.encoder.layers.0.attn.q_proj.bias", "clip.text.ln_final.weight"} <= keys
    enc.ctx = None


def test_host_philox_schedule_and_f16(dry):
    from mlimgsynth_amd import engine
    gold = json.load(open(os.path.join(GOLD, "reference_kats.json")))
    got, off = engine.randn(0, 0, 12)
    assert [f"{v:.8f}" for v in got] == gold["rng_seed0_offset0_n12"]          # src/test_rng.c:11-24
    got, _ = engine.randn(42, 0, 8)
    assert [f"{v:.8f}" for v in got] == gold["rng_seed42_offset0_n8"]
    assert off == 1
    for seed, o, n in [(42, 19, 4096), (2**64 - 1, 7, 33)]:
        assert np.array_equal(engine.randn(seed, o, n)[0].view(np.uint32), O.randn(seed, o, n).view(np.uint32))
    sig = engine.schedule("sd1", 20)
    assert [f"{v:.7g}" for v in sig] == [f"{v:.7g}" for v in gold["sigmas_20_uniform"]]
    sk = engine.schedule("sd1", 10, sched=2)                                     # Karras: vs the oracle's restatement
    ref = np.empty(32, np.float32)
    O.L().orc_schedule(10, 2, 1.0, 0.0, O.fptr(ref))
    assert np.array_equal(sk, ref[:11])
    # portable fp16 conversion of the weight loader == numpy RNE, incl. subnormals, ties, overflow
    l = engine.L()
    vals = np.concatenate([np.random.default_rng(0).standard_normal(2000).astype(np.float32) * s for s in (1e-7, 1e-4, 1, 300, 7e4)])
    vals = np.concatenate([vals, np.array([0, -0.0, 65504, 65519.99, 65520, 2**-24, 2**-25, 1.5 * 2**-24, 1 + 2**-11, 1 + 3 * 2**-11], np.float32)])
    with np.errstate(over="ignore"):
        exp = vals.astype(np.float16).view(np.uint16)
    got = np.array([l.mlb_f32_to_f16_bits(float(v)) for v in vals], np.uint16)
    assert np.array_equal(got, exp)
    back = np.array([l.mlb_f16_bits_to_f32(int(b)) for b in exp[:4000]], np.float32)
    assert np.array_equal(back, exp[:4000].view(np.float16).astype(np.float32))


def test_sdxl_label_layout(dry):
    from mlimgsynth_amd import text
    feat = np.arange(1280, dtype=np.float32)
    lab = text.sdxl_label(feat, 1024, 768)
    assert lab.shape == (2816,) and np.array_equal(lab[:1280], feat)
    # emb(h), emb(w) | emb(0), emb(0) | emb(h), emb(w); each [cos(256/2) | sin] (src/mlimgsynth.c:1485-1499,1548-1557)
    freq = np.exp(-np.log(10000.0) * np.arange(128) / 128)
    e = lambda v: np.concatenate([np.cos(v * freq), np.sin(v * freq)])
    exp = np.concatenate([e(768), e(1024), e(0), e(0), e(768), e(1024)])
    assert np.abs(lab[1280:] - exp).max() < 5e-4


def test_mlb_add_refuses_late_second_operand(dry):
    """mlb_add folds b into the epilogue of a's GEMM; a b recorded AFTER that GEMM (the reference's
    conv2 -> skip_conv -> add order, src/mlblock_nn.c:147-154) would be read before it is written.  It must fail
    loudly; b recorded first is accepted."""
    from mlimgsynth_amd import _lib
    L = dry.L()
    vp, c_int, c_bool = ctypes.c_void_p, ctypes.c_int, ctypes.c_bool
    L.mlctx_begin.argtypes = [vp, ctypes.c_char_p]
    L.mlctx_input_new.restype = vp
    L.mlctx_input_new.argtypes = [vp, ctypes.c_char_p, c_int, c_int, c_int, c_int, c_int]
    L.mlb_nn_conv2d.restype = vp
    L.mlb_nn_conv2d.argtypes = [vp, vp, c_int] + [c_int] * 8 + [c_bool]
    L.mlb_add.restype = vp
    L.mlb_add.argtypes = [vp, vp, vp]
    L.mlctx_tensor_add.restype = vp
    L.mlctx_tensor_add.argtypes = [vp, ctypes.c_char_p, vp]
    L.mlctx_prep.argtypes = [vp]

    def build(order):
        C = dry.MLCtx()
        L.mlctx_begin(C.h, b"addtest")
        x = L.mlctx_input_new(C.h, b"x", 0, 8, 8, 8, 1)
        if order == "late":
            a = L.mlctx_tensor_add(C.h, b"conv2", L.mlb_nn_conv2d(C.h, x, 16, 3, 3, 1, 1, 1, 1, 1, 1, True))
            b = L.mlctx_tensor_add(C.h, b"skip", L.mlb_nn_conv2d(C.h, x, 16, 1, 1, 1, 1, 0, 0, 1, 1, True))
        else:
            b = L.mlctx_tensor_add(C.h, b"skip", L.mlb_nn_conv2d(C.h, x, 16, 1, 1, 1, 1, 0, 0, 1, 1, True))
            a = L.mlctx_tensor_add(C.h, b"conv2", L.mlb_nn_conv2d(C.h, x, 16, 3, 3, 1, 1, 1, 1, 1, 1, True))
        y = L.mlb_add(C.h, a, b)
        if y:
            L.mlctx_tensor_add(C.h, b"out", y)
        return C, y, L.mlctx_prep(C.h)

    C, y, rc = build("late")
    assert not y and rc < 0 and "produced after" in _lib.last_error()
    C.destroy()
    C, y, rc = build("early")
    assert y and rc >= 1
    C.destroy()


def test_local_tensor_host_functions():
    """LocalTensor (src/localtensor.h:16-58,113-120, src/localtensor.c:63-69): resize owns memory and lets borrowed memory go,
    free resets, shape check treats n <= 0 as "any", finite check flags NaN / Inf.  Pure host code: runs without a GPU."""
    from mlimgsynth_amd import _lib
    L = _lib.lib()

    class LT(ctypes.Structure):
        _fields_ = [("d", ctypes.POINTER(ctypes.c_float)), ("n", ctypes.c_int * 4), ("flags", ctypes.c_int)]
    P = ctypes.POINTER(LT)
    L.ltensor_nelements.restype = ctypes.c_size_t; L.ltensor_nelements.argtypes = [P]
    L.ltensor_nbytes.restype = ctypes.c_size_t; L.ltensor_nbytes.argtypes = [P]
    L.ltensor_resize.restype = None; L.ltensor_resize.argtypes = [P] + [ctypes.c_int] * 4
    L.ltensor_free.restype = None; L.ltensor_free.argtypes = [P]
    L.ltensor_shape_check.argtypes = [P] + [ctypes.c_int] * 4
    L.ltensor_finite_check.argtypes = [P]
    borrowed = (ctypes.c_float * 6)(1, 2, 3, 4, 5, 6)
    t = LT(ctypes.cast(borrowed, ctypes.POINTER(ctypes.c_float)), (ctypes.c_int * 4)(3, 2, 1, 1), 0)
    assert L.ltensor_nelements(ctypes.byref(t)) == 6 and L.ltensor_nbytes(ctypes.byref(t)) == 24
    assert L.ltensor_shape_check(ctypes.byref(t), 3, 2, 0, -1) == 1 and L.ltensor_shape_check(ctypes.byref(t), 3, 3, 0, 0) == -1
    assert L.ltensor_finite_check(ctypes.byref(t)) == 1
    borrowed[4] = float("inf")
    assert L.ltensor_finite_check(ctypes.byref(t)) == -1
    L.ltensor_resize(ctypes.byref(t), 4, 4, 2, 1)                  # borrowed memory is not reallocated: a fresh owned block
    assert t.flags & 1 and list(t.n) == [4, 4, 2, 1] and ctypes.addressof(t.d.contents) != ctypes.addressof(borrowed)
    for i in range(32):
        t.d[i] = float(i)
    L.ltensor_resize(ctypes.byref(t), 8, 8, 1, 1)                  # owned memory grows in place (realloc): the old values survive
    assert [t.d[i] for i in range(32)] == [float(i) for i in range(32)]
    t.d[40] = float("nan")
    for i in list(range(32, 40)) + list(range(41, 64)):
        t.d[i] = 0.0
    assert L.ltensor_finite_check(ctypes.byref(t)) == -1
    L.ltensor_free(ctypes.byref(t))
    assert not t.d and list(t.n) == [0, 0, 0, 0] and t.flags == 0


def test_block_graph_dump_reproduces_the_parameter_names(dry, tmp_path):
    """mlctx_block_graph_dump_path (src/mlblock.c:347-388): the indented block tree, walked backwards like the name resolution.
    Joining every PARAM line with the names of its enclosing blocks must give exactly the parameter keys mlctx_prep derived
    (src/mlblock.c:67-105), with the same types and shapes; mlctx_build_alloc is the step-by-step name of prep."""
    L = dry.L()
    L.mlctx_block_graph_dump_path.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    for model in ("tinyxl", "sd1"):
        un = dry.Unet(model, 8, 8, 1, synth=False)
        path = str(tmp_path / f"{model}.txt")
        assert L.mlctx_block_graph_dump_path(un.ctx.h, path.encode()) == 1
        keys, stack = {}, []
        for line in open(path):
            depth = (len(line) - len(line.lstrip(" "))) // 2
            name, rest = line.strip().split(": ", 1)
            kind, typ, shape = rest.split(" ")
            del stack[depth:]
            if kind == "PARAM":
                keys[".".join(stack + [name])] = (typ, [int(v) for v in shape.strip("[]").split(",")])
            else:
                assert kind == "BLOCK"
                stack.append(name)
        want = {k: ("f16" if t == 1 else "f32", ne) for k, t, ne in un.ctx.param_list()}
        assert keys == want and len(keys) > 100
        un.ctx.destroy()
    # step-by-step interface name
    L.mlctx_build_alloc.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.mlctx_begin.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.mlctx_input_new.restype = ctypes.c_void_p
    L.mlctx_input_new.argtypes = [ctypes.c_void_p, ctypes.c_char_p] + [ctypes.c_int] * 5
    L.mlb_nn_linear.restype = ctypes.c_void_p
    L.mlb_nn_linear.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_bool]
    L.mlctx_tensor_add.restype = ctypes.c_void_p
    L.mlctx_tensor_add.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
    C = dry.MLCtx()
    L.mlctx_begin(C.h, b"ba")
    x = L.mlctx_input_new(C.h, b"x", 0, 64, 4, 1, 1)
    y = L.mlctx_tensor_add(C.h, b"fc", L.mlb_nn_linear(C.h, x, 32, True))
    assert L.mlctx_build_alloc(C.h, y) == 1
    assert [k for k, _, _ in C.param_list()] == ["fc.bias", "fc.weight"] or sorted(k for k, _, _ in C.param_list()) == ["fc.bias", "fc.weight"]
    C.destroy()


def test_weight_streaming_plan_in_the_dry_runtime(dry):
    """BASELINE configs[4] / the reference's --unet-split (src/unet.c:390-458): with mlctx_set_weight_streaming the plan's weights are cut into segments of consecutive
    ops that fit one of three slabs.  Host logic only (no GPU): every weight of the plan is streamed exactly once per evaluation except the step-invariant cross-attention
    K/V projections (resident), the segment count follows the slab size, a slab smaller than the largest single launch's weights is refused, and streaming + hipGraph
    replay is refused."""
    un = dry.Unet("tinyxl", 8, 8, 2, synth=False)
    assert un.ctx.streaming_info() is None
    total = sum(int(np.prod(ne)) * (2 if typ == 1 else 4) for _, typ, ne in un.ctx.param_list())
    infos = {}
    for mib in (1, 4):
        s = dry.Unet("tinyxl", 8, 8, 2, synth=False, stream_weights_mib=mib)
        nseg, per_eval, slab, host = s.ctx.streaming_info()
        infos[mib] = nseg
        assert slab == mib << 20 and nseg >= 2
        assert per_eval == host                                    # every streamed byte moves once per evaluation
        kv = sum(int(np.prod(ne)) * 2 for k, typ, ne in s.ctx.param_list() if ".attn2.k_proj." in k or ".attn2.v_proj." in k or "attn2.k_proj" in k or "attn2.v_proj" in k)
        assert 0.9 * (total - kv) <= host <= 1.1 * (total - kv) + 256 * len(s.ctx.param_list())      # (256-byte alignment per allocation)
        assert per_eval <= nseg * slab
        assert nseg <= s.ctx.streaming_copies() <= nseg + 2        # the host master is laid out in segment order: a segment's upload is one contiguous copy
        s.ctx.destroy()
    assert infos[1] > infos[4]
    with pytest.raises(Exception):
        dry.Unet("tinyxl", 8, 8, 2, synth=False, stream_weights_mib=1, flags=8)       # MLB_F_HIPGRAPH


@pytest.mark.parametrize("lw,lh", [(192, 128), (128, 192), (128, 128)])
def test_weight_streaming_prep_at_sizes_whose_dimensions_look_like_addresses(dry, lw, lh):
    """the prep-time scan for unpatched virtual weight addresses reads every 8-byte word of an op's arguments; a packed (n_img, HW) pair with
    HW = 192 x 128 = 24576 = 0x6000 read as the old virtual base 0x6000'0000'0000 and prep of a streamed plan at 1536 x 1024 failed.  The base is non-canonical now."""
    s = dry.Unet("sd1", lw, lh, 2, synth=False, stream_weights_mib=256)
    nseg, per_eval, slab, host = s.ctx.streaming_info()
    assert nseg >= 2 and per_eval == host
    s.ctx.destroy()
