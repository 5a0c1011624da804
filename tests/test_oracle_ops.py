"""Independent cross-check of the oracle's op restatements against torch CPU (fp64/fp32).
ggml itself is absent (parity unpinned, see oracle/oracle.h), so torch is the third party."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle_lib as O

torch.manual_seed(0)


def f16r(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def test_sgemm_nt_matches_numpy():
    rng = np.random.default_rng(0)
    for M, N, K in [(1, 1, 1), (7, 5, 3), (6, 16, 384), (130, 70, 777), (257, 300, 64), (33, 2049, 10)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
        C = np.empty((M, N), np.float32)
        O.L().orc_sgemm_nt(M, N, K, O.fptr(A), K, O.fptr(B), K, O.fptr(C), N)
        assert rel(C, A.astype(np.float64) @ B.astype(np.float64).T) < 2e-6, (M, N, K)


def test_sgemm_micro_kernels_give_the_same_bits():
    """AVX2 6x16 and AVX-512 12x32 micro-kernels (run-time choice, oracle/o_core.c): every element is FMA-accumulated in k order inside the same K chunks, so
    the vector width, the tile sizes and the thread count do not change a bit.  Ragged shapes, K across chunk borders, strided C."""
    L = O.L()
    rng = np.random.default_rng(3)
    L.orc_set_isa(5)
    if L.orc_get_isa() != 5:
        pytest.skip("no AVX-512 on this CPU")
    try:
        for M, N, K in [(1, 1, 1), (13, 33, 385), (130, 70, 777), (257, 300, 1153), (6, 2049, 64), (640, 31, 2880)]:
            A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
            out = {}
            for isa, th in ((2, 0), (5, 0), (5, 1), (2, 3)):
                L.orc_set_isa(isa)
                if th:
                    L.orc_set_threads(th)
                C = np.full((M, N + 5), np.float32(7.0)); L.orc_sgemm_nt(M, N, K, O.fptr(A), K, O.fptr(B), K, O.fptr(C), N + 5)
                assert np.all(C[:, N:] == 7.0)
                out[(isa, th)] = C[:, :N].copy()
            ref = out[(2, 0)]
            assert all(np.array_equal(ref, v) for v in out.values()), (M, N, K)
    finally:
        L.orc_set_isa(-1); L.orc_set_threads(O.host_threads())


def test_round_f16_is_rne():
    x = np.array([1.0, 1.0 + 2**-11, 1.0 + 3 * 2**-11, 65504.0, 1e-8, -2.5, 0.1, 70000.0], np.float32)
    y = x.copy()
    O.L().orc_round_f16(O.fptr(y), y.size)
    with np.errstate(over="ignore"):
        assert np.array_equal(y, x.astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("cin,cout,k,s,p,h,w", [(8, 16, 3, 1, 1, 9, 7), (32, 8, 3, 2, 1, 8, 8), (16, 32, 1, 1, 0, 5, 6),
                                                  (4, 64, 3, 1, 1, 8, 8), (64, 3, 3, 1, 1, 6, 6)])
def test_conv2d(cin, cout, k, s, p, h, w):
    rng = np.random.default_rng(1)
    x = f16r(rng.standard_normal((1, cin, h, w)))
    wt = rng.standard_normal((cout, cin, k, k)).astype(np.float32) / np.sqrt(cin * k * k)
    b = rng.standard_normal(cout).astype(np.float32)
    P = O.Params()
    pw, pb = P.set("w", wt, f16=True), P.set("b", b)
    y = O.from_ot(O.L().orc_conv2d(O.to_ot(x), pw, pb, s, p))
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(f16r(wt)).double(), torch.from_numpy(b).double(),
                   stride=s, padding=p).numpy()
    assert y.shape == ref.shape
    assert rel(y, ref) < 2e-6


def test_conv2d_rounds_activations_to_f16():
    """ggml im2col is F16: an activation that is not f16-representable is rounded first."""
    x = np.full((1, 1, 1, 1), 1.0 + 2**-12, np.float32)
    P = O.Params()
    pw = P.set("w", np.ones((1, 1, 1, 1), np.float32), f16=True)
    y = O.from_ot(O.L().orc_conv2d(O.to_ot(x), pw, None, 1, 0))
    assert y.item() == 1.0


def test_vae_downsample_pad_end():
    rng = np.random.default_rng(2)
    x = f16r(rng.standard_normal((1, 8, 8, 8)))
    wt = f16r(rng.standard_normal((8, 8, 3, 3)) / 8)
    P = O.Params()
    pw = P.set("w", wt, f16=True)
    xp = O.L().orc_pad_end(O.to_ot(x), 1, 1)
    y = O.from_ot(O.L().orc_conv2d(xp, pw, None, 2, 0))
    ref = F.conv2d(F.pad(torch.from_numpy(x), (0, 1, 0, 1)), torch.from_numpy(wt), stride=2).numpy()
    assert y.shape == ref.shape == (1, 8, 4, 4)
    assert rel(y, ref) < 2e-6


def test_linear_f16_and_f32_weights():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 1, 13, 96)).astype(np.float32)
    wt = rng.standard_normal((40, 96)).astype(np.float32) / 10
    b = rng.standard_normal(40).astype(np.float32)
    P = O.Params()
    y16 = O.from_ot(O.L().orc_linear(O.to_ot(x), P.set("w16", wt, f16=True), P.set("b", b)))
    y32 = O.from_ot(O.L().orc_linear(O.to_ot(x), P.set("w32", wt, f16=False), None))
    ref16 = f16r(x).reshape(13, 96).astype(np.float64) @ f16r(wt).astype(np.float64).T + b
    ref32 = x.reshape(13, 96).astype(np.float64) @ wt.astype(np.float64).T
    assert rel(y16.reshape(13, 40), ref16) < 2e-6
    assert rel(y32.reshape(13, 40), ref32) < 2e-6


def test_group_norm_32_eps1e6():
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((1, 64, 6, 5)) * 3 + 1.5).astype(np.float32)
    g, b = rng.standard_normal(64).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    P = O.Params()
    y = O.from_ot(O.L().orc_group_norm(O.to_ot(x), 32, 1e-6, P.set("g", g), P.set("b", b)))
    ref = F.group_norm(torch.from_numpy(x).double(), 32, torch.from_numpy(g).double(), torch.from_numpy(b).double(),
                       eps=1e-6).numpy()
    assert rel(y, ref) < 2e-6


def test_layer_norm():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((1, 1, 11, 320)) * 2 - 0.7).astype(np.float32)
    g, b = rng.standard_normal(320).astype(np.float32), rng.standard_normal(320).astype(np.float32)
    P = O.Params()
    y = O.from_ot(O.L().orc_layer_norm(O.to_ot(x), 0.0, P.set("g", g), P.set("b", b)))
    ref = F.layer_norm(torch.from_numpy(x).double(), (320,), torch.from_numpy(g).double(),
                       torch.from_numpy(b).double(), eps=1e-5).numpy()
    assert rel(y, ref) < 2e-6


@pytest.mark.parametrize("tq,tk,heads,dh,causal", [(17, 17, 2, 40, False), (9, 77, 4, 64, False), (77, 77, 3, 64, True)])
def test_attention(tq, tk, heads, dh, causal):
    rng = np.random.default_rng(6)
    D = heads * dh
    q = rng.standard_normal((1, 1, tq, D)).astype(np.float32)
    k = rng.standard_normal((1, 1, tk, D)).astype(np.float32)
    v = rng.standard_normal((1, 1, tk, D)).astype(np.float32)
    y = O.from_ot(O.L().orc_attention(O.to_ot(q), O.to_ot(k), O.to_ot(v), heads, int(causal))).reshape(tq, D)
    tq_, tk_, tv_ = (torch.from_numpy(a.reshape(-1, heads, dh)).double().transpose(0, 1) for a in (q, k, v))
    ref = F.scaled_dot_product_attention(tq_, tk_, tv_, is_causal=causal).transpose(0, 1).reshape(tq, D).numpy()
    assert rel(y, ref) < 2e-6


def test_activations_and_small_ops():
    x = np.linspace(-6, 6, 101, dtype=np.float32).reshape(1, 1, 1, 101)
    L = O.L()
    for fn, ref in [("orc_silu", F.silu), ("orc_gelu", lambda t: F.gelu(t, approximate="tanh")),
                    ("orc_gelu_quick", lambda t: t * torch.sigmoid(1.702 * t)), ("orc_relu", F.relu)]:
        t = O.to_ot(x)
        getattr(L, fn)(t)
        assert rel(O.from_ot(t), ref(torch.from_numpy(x).double()).numpy()) < 1e-6, fn
    u = O.from_ot(L.orc_upscale2(O.to_ot(np.arange(6, dtype=np.float32).reshape(1, 1, 2, 3))))
    assert np.array_equal(u, F.interpolate(torch.arange(6.).reshape(1, 1, 2, 3), scale_factor=2, mode="nearest").numpy())
    # timestep embedding: cos first, then sin (src/mlimgsynth.c:1485-1499)
    t = np.array([999.0, 3.5], np.float32)
    out = np.empty((2, 320), np.float32)
    L.orc_timestep_embedding(O.fptr(t), 2, 320, 10000.0, O.fptr(out))
    freq = np.exp(-np.log(10000.0) * np.arange(160) / 160)
    ref = np.concatenate([np.cos(t[:, None] * freq), np.sin(t[:, None] * freq)], 1)
    assert np.abs(out - ref).max() < 2e-4


def test_ggml_f16_table_mode_of_the_activations():
    """orc_set_ggml_f16_tables(1): GELU / quick-GELU as ggml's CPU backend evaluates them -- through F16 lookup tables, i.e. fp16(x) -> formula -> fp16
    (GELU passes x <= -10 / x >= 10 through exactly).  Held against a numpy restatement of the table construction; the default mode stays exact."""
    L = O.L()
    x = np.concatenate([np.linspace(-12, 12, 4001), [1e-4, -1e-4, 3.14159, 65504.0, -70000.0]]).astype(np.float32).reshape(1, 1, 1, -1)
    xs = x.ravel()
    xh = xs.astype(np.float16).astype(np.float32)
    gelu = lambda v: (0.5 * v * (1.0 + np.tanh(np.float32(0.7978845608028654) * v * (1.0 + np.float32(0.044715) * v * v)))).astype(np.float32)
    quick = lambda v: (v / (1.0 + np.exp(np.float32(-1.702) * v))).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        want_g = np.where(xs <= -10, 0.0, np.where(xs >= 10, xs, gelu(xh).astype(np.float16).astype(np.float32))).astype(np.float32)
        want_q = quick(xh).astype(np.float16).astype(np.float32)
    try:
        L.orc_set_ggml_f16_tables(1)
        assert L.orc_get_ggml_f16_tables() == 1
        for fn, want in (("orc_gelu", want_g), ("orc_gelu_quick", want_q)):
            t = O.to_ot(x)
            getattr(L, fn)(t)
            got = O.from_ot(t).ravel()
            fin = np.isfinite(want)
            # (tanhf / expf of libm against numpy: the same value before the final fp16 rounding up to 1 ulp, so at most one fp16 step apart, on a few points)
            assert np.array_equal(np.isfinite(got), fin), fn
            step = np.spacing(np.abs(want[fin]).astype(np.float16)).astype(np.float32)
            # (around x = -5.4 the fp32 formula cancels, 1 + tanh(..) ~ 2e-8, and libm's tanhf and numpy's differ there by a few fp16 subnormal steps of the result)
            assert (np.abs(got[fin] - want[fin]) <= np.maximum(step, 2.5e-7)).all(), fn
            assert (got[fin] == want[fin]).mean() > 0.95, fn
            assert np.array_equal(got[fin], got[fin].astype(np.float16).astype(np.float32)) or fn == "orc_gelu", fn   # table outputs are fp16 values
    finally:
        L.orc_set_ggml_f16_tables(0)
    t = O.to_ot(x[..., :4001])
    L.orc_gelu(t)
    assert rel(O.from_ot(t), F.gelu(torch.from_numpy(x[..., :4001]).double(), approximate="tanh").numpy()) < 1e-6     # default: exact formula


def test_unet_tiny_runs_and_names_follow_reference():
    U = O.unet_params("tiny")
    P = O.Params(1234)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    ctx = rng.standard_normal((1, 1, 77, U.n_ctx)).astype(np.float32)
    y = O.from_ot(O.L().orc_unet_graph(P.h, b"unet", U, O.to_ot(x), 500.0, O.to_ot(ctx), None))
    assert y.shape == (1, 4, 8, 8) and np.isfinite(y).all() and y.std() > 0.01
    names = {n for n, _, _ in P.names()}
    # dotted keys as derived by mlctx_load_prep (src/mlblock.c:67-105)
    for key in ["unet.time_embed.0.weight", "unet.in.conv.weight", "unet.in.1.0.norm1.weight",
                "unet.in.1.0.emb_proj.bias", "unet.in.1.1.transf.0.attn1.q_proj.weight",
                "unet.in.1.1.transf.0.ff.net.0.proj.bias", "unet.in.2.0.conv.weight", "unet.mid.1.proj_out.weight",
                "unet.out.0.0.skip_conv.weight", "unet.out.1.2.conv.weight", "unet.out.norm.bias", "unet.out.conv.weight"]:
        assert key in names, key
    # determinism of the synthetic weights
    P2 = O.Params(1234)
    y2 = O.from_ot(O.L().orc_unet_graph(P2.h, b"unet", U, O.to_ot(x), 500.0, O.to_ot(ctx), None))
    assert np.array_equal(y, y2)


def test_linear_weight_type_follows_the_checkpoint():
    """The reference takes the type of the Linear weights from the checkpoint (src/mlimgsynth.c:1235-1236, src/mlblock_nn.c:20-22): an fp32 checkpoint (BASELINE configs[0]) makes
    ggml multiply fp32 weights with fp32 activations, an fp16 one rounds both to F16.  The oracle restates both (orc_set_linear_wtype); the device always holds F16 weights
    (DESIGN.md section 8), i.e. it computes the F16 form from an fp32 checkpoint too.  That deviation is MEASURED here on the tiny model and bounded by the per-evaluation
    tolerance; tools/fp32_checkpoint_deviation.py has it for SD1.5 / SDXL at latent 32: 1.3e-3 / 1.4e-3 per evaluation, 2.5e-3 / 4.6e-3 on the final latent of 20 steps."""
    import ctypes
    import tolerances as T
    L = O.L()
    L.orc_set_linear_wtype.argtypes = [ctypes.c_int]
    U = O.unet_params("tiny")
    rng = np.random.default_rng(11)
    x = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    ctx = rng.standard_normal((1, 1, 77, U.n_ctx)).astype(np.float32)
    ys = {}
    try:
        for wt in (1, 0):
            L.orc_set_linear_wtype(wt)
            assert L.orc_get_linear_wtype() == wt
            P = O.Params(1234)
            ys[wt] = O.from_ot(L.orc_unet_graph(P.h, b"unet", U, O.to_ot(x), 500.0, O.to_ot(ctx), None))
            types = {n: t for n, t, _ in P.names()}
            assert types["unet.in.1.1.transf.0.attn1.q_proj.weight"] == wt and types["unet.time_embed.0.weight"] == wt      # Linear: the checkpoint's type
            assert types["unet.in.conv.weight"] == 1                                                                        # Conv2d: F16 always (src/mlblock_nn.c:42-43)
            P.free()
    finally:
        L.orc_set_linear_wtype(1)
    e = rel(ys[1], ys[0])
    print(f"tiny UNet, F16 against F32 linear weights: rel-L2 {e:.2e}")
    assert 1e-5 < e < T.EVAL


def test_softmax_exponential_of_the_oracle_against_expf():
    """the oracle -- the ground truth of the attention parity tests -- computes its softmax exponential with an 8-lane degree-7 polynomial instead of expf
    (tail lanes of a row still use expf): held here against float64 exp over the whole input range [-87, 0], on -inf (masked keys) and below the cut-off, bound 2e-7 relative."""
    import ctypes
    L = O.L()
    L.orc_exp_sub.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int64, ctypes.c_float]
    L.orc_exp_sub.restype = None
    rng = np.random.default_rng(0)
    for mx in (0.0, 3.25, -11.5):
        t = np.concatenate([np.linspace(-87.0, 0.0, 200001), -rng.random(50000) * 87.0, -rng.random(50000) ** 4 * 5.0, [0.0, -1e-8, -86.99]]).astype(np.float32)
        t = t[: (t.size // 8) * 8 + 5]                                   # 5 tail lanes: the scalar expf path of the same row
        x = (t + np.float32(mx)).astype(np.float32)
        want = np.exp((x.astype(np.float64) - np.float64(np.float32(mx))))
        got = x.copy()
        L.orc_exp_sub(got.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), got.size, mx)
        live = (x - np.float32(mx)) >= -87.0
        rel = np.abs(got[live].astype(np.float64) - want[live]) / want[live]
        assert rel.max() < 2e-7, (mx, rel.max())
        assert np.all(got[~live][: (got.size // 8) * 8] >= 0) and np.all(got[~live] < 2e-38)
    dead = np.array([-np.inf, -1000.0, -88.0, -87.5, -np.inf, -200.0, -np.inf, -90.0, 0.0], np.float32)     # one 8-lane group of dead keys + a live tail lane
    L.orc_exp_sub(dead.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), dead.size, 0.0)
    assert np.array_equal(dead[:8], np.zeros(8, np.float32)) and dead[8] == 1.0
    # a softmax row is sum-normalised afterwards: the polynomial and expf lanes of one row agree to the same bound
    row = (-rng.random(77) * 30).astype(np.float32)
    a = row.copy(); L.orc_exp_sub(a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.size, 0.0)
    b = np.exp(row.astype(np.float64))
    assert np.abs(a / a.sum() - b / b.sum()).max() / (b / b.sum()).max() < 3e-7
