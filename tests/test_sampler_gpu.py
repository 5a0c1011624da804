"""The general sampler of the engine (all solvers of src/solvers.c, Karras schedule, stochastic noise, v-prediction,
img2img, in-painting: SURVEY.md section 8 rows f2/f3) against the oracle's restatement of dnsamp_step, and the device vector
kernels against numpy restatements of the reference's loops (bit-exact: same fp32/fp64 operation order)."""
import ctypes

import numpy as np
import tolerances as T
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
F = np.float32


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def dev(_lib, a):
    return _lib.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def K():
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    i64, cf, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
    L.mlsd_vec_axpy.argtypes = [vp, vp, vp, cf, i64, vp]
    L.mlsd_solver_heun_corr.argtypes = [vp, vp, vp, cf, i64, vp]
    L.mlsd_solver_taylor3.argtypes = [vp, vp, vp, vp, cf, cf, cf, cf, i64, vp]
    L.mlsd_solver_dpmpp2m.argtypes = [vp, vp, vp, cf, cf, cf, i64, vp]
    L.mlsd_solver_dpmpp2s.argtypes = [vp, vp, vp, cf, cf, i64, vp]
    L.mlsd_noise_add_s.argtypes = [vp, vp, cf, i64, vp]
    L.mlsd_mask_apply.argtypes = [vp, vp, vp, ctypes.c_int, i64, vp]
    L.mlsd_dxdt_cfg.argtypes = [vp, i64, vp, vp] + [ctypes.c_int] * 3 + [cf, ctypes.c_int, cf, cf, vp]
    L.mlsd_euler_cfg_update.argtypes = [vp, vp, i64] + [ctypes.c_int] * 3 + [cf, cf, vp, cf, vp]
    L.mlsd_latent_sample.argtypes = [vp, i64, vp, vp] + [ctypes.c_int] * 3 + [cf, vp]
    return _lib, L


def test_solver_kernels_bit_exact_vs_reference_loops(K):
    _lib, L = K
    r = np.random.default_rng(5)
    n = 4 * 4 * 37
    x, dx, a, b = [r.standard_normal(n).astype(F) for _ in range(4)]
    dt, t0 = F(-1.37), F(3.3)
    # euler / predictor  x1 = x + dx*dt  (solvers.c:86,105)
    d = [dev(_lib, v) for v in (x, dx, a, b)]
    out = _lib.DeviceBuffer(n * 4)
    _lib.check(L.mlsd_vec_axpy(out.ptr, d[0].ptr, d[1].ptr, dt, n, None))
    assert np.array_equal(out.download((n,), F), x + dx * dt)
    # heun corrector (solvers.c:112-113): (dx + d1) * 0.5 * dt in double, added to x in double
    xx = dev(_lib, x)
    _lib.check(L.mlsd_solver_heun_corr(xx.ptr, d[1].ptr, d[2].ptr, dt, n, None))
    ref = (x.astype(np.float64) + (dx + a).astype(np.float64) * 0.5 * np.float64(dt)).astype(F)
    assert np.array_equal(xx.download((n,), F), ref)
    # taylor3 (solvers.c:150-165)
    idtp, f2, f3 = F(1) / F(-1.1), dt * dt / F(2), dt * dt * dt / F(6)
    xx, dp1, dp2 = dev(_lib, x), dev(_lib, a), dev(_lib, b)
    _lib.check(L.mlsd_solver_taylor3(xx.ptr, d[1].ptr, dp1.ptr, dp2.ptr, dt, idtp, f2, f3, n, None))
    x1 = x + dx * dt
    d2 = (dx - a) * idtp
    d3 = (d2 - b) * idtp
    assert np.array_equal(xx.download((n,), F), x1 + (d2 * f2 + d3 * f3))
    assert np.array_equal(dp1.download((n,), F), dx) and np.array_equal(dp2.download((n,), F), d2)
    # dpmpp2m (solvers.c:222-229)
    aa, c = F(0.8), F(0.45)
    xx, dprev = dev(_lib, x), dev(_lib, a)
    _lib.check(L.mlsd_solver_dpmpp2m(xx.ptr, d[1].ptr, dprev.ptr, t0, aa, c, n, None))
    d0 = x - t0 * dx
    dd = (F(1) + c) * d0 - c * a
    assert np.array_equal(xx.download((n,), F), aa * x + (F(1) - aa) * dd)
    assert np.array_equal(dprev.download((n,), F), d0)
    # dpmpp2s second half (solvers.c:281-284)
    xx = dev(_lib, x)
    _lib.check(L.mlsd_solver_dpmpp2s(xx.ptr, d[2].ptr, d[3].ptr, t0, aa, n, None))
    dd = a - t0 * b
    assert np.array_equal(xx.download((n,), F), aa * x + (F(1) - aa) * dd)
    # noise add, mask blend (sampling.c:98-117)
    xx = dev(_lib, x)
    _lib.check(L.mlsd_noise_add_s(xx.ptr, d[2].ptr, F(0.7), n, None))
    assert np.array_equal(xx.download((n,), F), x + a * F(0.7))
    m = r.random(37).astype(F)
    xx, md = dev(_lib, x), dev(_lib, m)
    _lib.check(L.mlsd_mask_apply(xx.ptr, d[3].ptr, md.ptr, 37, n, None))
    mm = np.tile(m, 16)
    assert np.array_equal(xx.download((n,), F), b * mm + x * (F(1) - mm))


def test_dxdt_cfg_vparam_and_fused_euler_bit_exact(K):
    _lib, L = K
    r = np.random.default_rng(6)
    B, C, HW = 2, 4, 33
    x = r.standard_normal((B, C, HW)).astype(F)
    eps = r.standard_normal((2 * B, HW, C)).astype(F)
    noise = r.standard_normal((B, C, HW)).astype(F)
    f, c_out, c_skip = F(7), F(0.31), F(0.29)
    ec, eu = eps[:B].transpose(0, 2, 1), eps[B:].transpose(0, 2, 1)
    de, dx_, dxb = dev(_lib, eps), dev(_lib, x), _lib.DeviceBuffer(x.nbytes)
    _lib.check(L.mlsd_dxdt_cfg(de.ptr, C, dx_.ptr, dxb.ptr, B, C, HW, f, 0, c_out, c_skip, None))
    assert np.array_equal(dxb.download(x.shape, F), ec * f + eu * (F(1) - f))                       # mlimgsynth.c:1583
    _lib.check(L.mlsd_dxdt_cfg(de.ptr, C, dx_.ptr, dxb.ptr, B, C, HW, f, 1, c_out, c_skip, None))
    vc, vu = ec * c_out + x * c_skip, eu * c_out + x * c_skip                                      # unet.c:493
    assert np.array_equal(dxb.download(x.shape, F), vc * f + vu * (F(1) - f))
    # fused == dxdt_cfg + axpy + noise add
    dt, sup = F(-0.9), F(0.4)
    xx, dn = dev(_lib, x), dev(_lib, noise)
    _lib.check(L.mlsd_euler_cfg_update(xx.ptr, de.ptr, C, B, C, HW, f, dt, dn.ptr, sup, None))
    ref = x + (ec * f + eu * (F(1) - f)) * dt
    assert np.array_equal(xx.download(x.shape, F), ref + noise * sup)


def test_latent_sample_matches_oracle(K):
    _lib, L = K
    r = np.random.default_rng(7)
    B, cz, HW = 2, 4, 64
    mom = (r.standard_normal((B, HW, 2 * cz)) * 3).astype(F)           # NHWC as the encoder plan leaves it
    mom[0, 0, cz] = 40; mom[0, 1, cz] = -50                            # clamp(logvar, -30, 20)
    rnd = r.standard_normal((B, cz, HW)).astype(F)
    out = _lib.DeviceBuffer(B * cz * HW * 4)
    V = O.vae_params("sd1")
    for use_rnd in (True, False):
        dm, dr = dev(_lib, mom), dev(_lib, rnd)
        _lib.check(L.mlsd_latent_sample(dm.ptr, 2 * cz, dr.ptr if use_rnd else None, out.ptr, B, cz, HW, V.scale_factor, None))
        got = out.download((B, cz, 8, 8), F)
        for b in range(B):
            m_nchw = mom[b].T.reshape(1, 2 * cz, 8, 8)
            ref = O.from_ot(O.L().orc_latent_sample(O.to_ot(m_nchw), V, O.fptr(np.ascontiguousarray(rnd[b])) if use_rnd else None))[0]
            assert np.allclose(got[b], ref, rtol=2e-6, atol=1e-7)       # exp(): device libm vs glibc, otherwise the same ops


CASES = [   # method, s_ancestral, sched, s_noise, steps
    ("euler", 0.0, 1, 0.0, 8), ("euler", 1.0, 2, 0.0, 8), ("euler", 0.0, 1, 0.8, 6),
    ("heun", 0.0, 1, 0.0, 8), ("heun", 1.0, 2, 0.0, 8),
    ("taylor3", 0.0, 1, 0.0, 8), ("taylor3", 1.0, 1, 0.0, 6),
    ("dpmpp2m", 0.0, 2, 0.0, 8), ("dpmpp2m", 1.0, 1, 0.0, 8),
    ("dpmpp2s", 1.0, 1, 0.0, 8), ("dpmpp2s", 0.0, 2, 0.0, 9),
]
METHOD_ID = {"euler": 1, "heun": 2, "taylor3": 3, "dpmpp2m": 4, "dpmpp2s": 5}


def gen_inputs(model, seed=8):
    U = O.unet_params(model)
    rng = np.random.default_rng(seed)
    cond = rng.standard_normal((77, U.n_ctx)).astype(F)
    uncond = rng.standard_normal((77, U.n_ctx)).astype(F)
    label = rng.standard_normal(U.ch_adm_in).astype(F) if U.ch_adm_in else None
    unlabel = rng.standard_normal(U.ch_adm_in).astype(F) if U.ch_adm_in else None
    return U, cond, uncond, label, unlabel


def oracle_sample(model, lat, cond, uncond, label, unlabel, method, s_anc, sched, s_noise, steps, seed, cfg=7.0, f_t_ini=1.0,
                  init=None, lmask=None, rng_offset=0):
    U = O.unet_params(model)
    P = O.Params(1234)
    opts = O.SampleOpts(METHOD_ID[method], sched, steps, cfg, s_anc, s_noise, f_t_ini, 0.0)
    out = np.empty((4, lat, lat), F)
    nfe = O.L().orc_sample_ex(P.h, b"unet", U, lat, lat, O.to_ot(cond[None, None]),
                              O.to_ot(label[None, None, None]) if label is not None else None, O.to_ot(uncond[None, None]),
                              O.to_ot(unlabel[None, None, None]) if unlabel is not None else None, ctypes.byref(opts), seed, rng_offset,
                              O.fptr(np.ascontiguousarray(init)) if init is not None else None,
                              O.fptr(np.ascontiguousarray(lmask)) if lmask is not None else None, O.fptr(out))
    P.free()
    return out, nfe


@pytest.mark.parametrize("method,s_anc,sched,s_noise,steps", CASES)
def test_solvers_and_schedulers_vs_oracle(method, s_anc, sched, s_noise, steps):
    from mlimgsynth_amd import engine
    model, lat, B = "tiny", 8, 2
    U, cond, uncond, label, unlabel = gen_inputs(model)
    g = engine.Generator(model, lat * 8, lat * 8, B, n_step=steps, cfg_scale=7.0, s_ancestral=s_anc, sched=sched, method=method,
                         s_noise=s_noise)
    g.set_cond(cond, label, uncond, unlabel)
    seeds = [11, 12]
    got, _ = g.generate(seeds, want_images=False)
    # (the CPU oracle is most of this test's time: both images of the batch for the Euler cases, image 1 -- the slot that is not first -- for the other solvers)
    for b in (range(B) if method == "euler" else (1,)):
        ref, nfe = oracle_sample(model, lat, cond, uncond, label, unlabel, method, s_anc, sched, s_noise, steps, seeds[b])
        assert g.last_nfe() == nfe
        e = rel(got[b], ref)
        print(method, s_anc, sched, s_noise, "image", b, "rel-L2", e, "nfe", nfe)
        assert np.isfinite(got[b]).all() and e < T.LATENT
    g.destroy()


def test_vparam_generation_vs_oracle():
    """an accepted v-prediction config must apply out*c_out + x*c_skip (src/unet.c:490-494) in the driver"""
    from mlimgsynth_amd import engine
    model, lat = "tinyv", 8
    U, cond, uncond, label, unlabel = gen_inputs(model)
    assert U.vparam == 1
    for method in ("euler", "dpmpp2m"):
        g = engine.Generator(model, lat * 8, lat * 8, 1, n_step=8, cfg_scale=5.0, s_ancestral=1.0, method=method)
        g.set_cond(cond, None, uncond, None)
        got, _ = g.generate([21], want_images=False)
        ref, nfe = oracle_sample(model, lat, cond, uncond, None, None, method, 1.0, 1, 0.0, 8, 21, cfg=5.0)
        e = rel(got[0], ref)
        print("vparam", method, e)
        assert g.last_nfe() == nfe and e < T.LATENT
        # and it is NOT what an eps-model driver would return
        g.destroy()


def test_cfg_off_single_evaluation_per_step():
    from mlimgsynth_amd import engine
    U, cond, uncond, _, _ = gen_inputs("tiny")
    g = engine.Generator("tiny", 64, 64, 1, n_step=5, cfg_scale=1.0, s_ancestral=0.0)
    g.set_cond(cond, None, None, None)
    got, _ = g.generate([3], want_images=False)
    ref, nfe = oracle_sample("tiny", 8, cond, uncond, None, None, "euler", 0.0, 1, 0.0, 5, 3, cfg=1.0)
    assert g.last_nfe() == nfe == 5 and rel(got[0], ref) < T.LATENT


def test_img2img_and_inpaint_vs_oracle():
    """img2img: initial latent + f_t_ini < 1 (fewer steps, lower start sigma); in-painting: latent mask blend after every
    noise add and step (src/sampling.c:54-56,98-110,129-136,176-178)."""
    from mlimgsynth_amd import engine
    model, lat = "tiny", 8
    U, cond, uncond, _, _ = gen_inputs(model)
    rng = np.random.default_rng(9)
    init = (rng.standard_normal((1, 4, lat, lat)) * 0.8).astype(F)
    lmask = (rng.random((lat, lat)) > 0.5).astype(F)
    lmask[0, 0] = 0.25
    g = engine.Generator(model, lat * 8, lat * 8, 1, n_step=10, cfg_scale=7.0, s_ancestral=1.0, f_t_ini=0.6)
    g.set_cond(cond, None, uncond, None)
    for mask in (None, lmask):
        g.set_init_latent(init)
        g.set_lmask(mask)
        got, _ = g.generate([5], want_images=False)
        ref, nfe = oracle_sample(model, lat, cond, uncond, None, None, "euler", 1.0, 1, 0.0, 10, 5, f_t_ini=0.6, init=init[0], lmask=mask)
        assert g.last_n_step() == 6 and g.last_nfe() == nfe == 12
        e = rel(got[0], ref)
        print("img2img", "mask" if mask is not None else "nomask", e)
        assert e < T.LATENT
        if mask is not None:    # fully kept pixels equal the original latent exactly
            keep = np.broadcast_to(lmask == 1, got[0].shape)
            assert np.array_equal(got[0][keep], init[0][keep])
    g.destroy()


def test_progress_callback_and_abort():
    from mlimgsynth_amd import engine
    U, cond, uncond, _, _ = gen_inputs("tiny")
    g = engine.Generator("tiny", 64, 64, 1, n_step=6)
    g.set_cond(cond, None, uncond, None)
    calls = []
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int)

    def cb(user, step, n_step, nfe):
        calls.append((step, n_step, nfe))
        return -42 if step == 3 else 1
    cbc = CB(cb)
    engine._proto2().mlis_amd_set_callback(g.h, ctypes.cast(cbc, ctypes.c_void_p), None)
    from mlimgsynth_amd import _lib
    with pytest.raises(_lib.MlsdError):
        g.generate([1])
    assert calls == [(1, 6, 2), (2, 6, 4), (3, 6, 6)]
    # (1) the abort leaves the Philox state at the draws CONSUMED (the loop generates one step ahead): 1 initial + 3 ancestral draws were added to the latent when
    # step 3 reported; (2) a callback that aborts with -8 -- the value the loop once used in band for "hand-off timed out, run again" -- aborts, it does not re-run the loop
    l2 = engine._proto2()
    l2.mlis_amd_rng_offset.argtypes = [ctypes.c_void_p]; l2.mlis_amd_rng_offset.restype = ctypes.c_uint32
    l2.mlis_amd_handoff_retries.argtypes = [ctypes.c_void_p]
    assert l2.mlis_amd_rng_offset(g.h) == 4
    calls.clear()

    def cb8(user, step, n_step, nfe):
        calls.append(step)
        return -8 if step == 2 else 1
    cbc8 = CB(cb8)
    l2.mlis_amd_set_callback(g.h, ctypes.cast(cbc8, ctypes.c_void_p), None)
    with pytest.raises(_lib.MlsdError):
        g.generate([1])
    assert calls == [1, 2] and l2.mlis_amd_handoff_retries(g.h) == 0
    assert l2.mlis_amd_rng_offset(g.h) == 3


@pytest.mark.parametrize("model,side", [("tiny", 64), ("sd1", 64)])
def test_vae_encoder_vs_oracle_and_golden(model, side):
    """KL-VAE encoder (end-padded stride-2 convs, src/mlblock_nn.c:105-116) moments vs the oracle and the independent fixture"""
    import os
    import golden_cases as G
    from mlimgsynth_amd import engine
    l = engine._proto2()
    key = f"vaeenc_{model}_{side}"
    img = G.image_inputs(key, side)
    ctx = engine.MLCtx()
    P = engine.VaeParams()
    l.vae_params_get(model.encode(), ctypes.byref(P))
    t_img = engine.vp()
    l.sdvae_encode_init.argtypes = [engine.vp, ctypes.POINTER(engine.VaeParams), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(engine.vp)]
    l.sdvae_encode_build.argtypes = [engine.vp, ctypes.POINTER(engine.VaeParams), engine.vp]
    l.sdvae_encode_run.argtypes = [engine.vp, engine.vp, engine.FP, engine.FP]
    engine.check1(l.sdvae_encode_init(ctx.h, ctypes.byref(P), side, side, 1, ctypes.byref(t_img)), "init")
    engine.check1(l.sdvae_encode_build(ctx.h, ctypes.byref(P), t_img), "build")
    ctx.params_synth(1234)
    mom = np.empty((1, 8, side // 8, side // 8), F)
    engine.check1(l.sdvae_encode_run(ctx.h, t_img, engine.fptr(img), engine.fptr(mom)), "run")
    OP = O.Params(1234)
    ref = O.from_ot(O.L().orc_vae_encode_moments(OP.h, b"vae", O.vae_params(model), O.to_ot(img)))
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_golden.npz"))[key]
    print(key, "vs oracle", rel(mom, ref), "vs independent golden", rel(mom, gold))
    assert rel(mom, ref) < T.EVAL and rel(mom, gold) < T.EVAL
    assert {k for k, _, _ in ctx.param_list()} == {k for k, _, _ in OP.names()}


def test_engine_encode_sample_and_img2img_roundtrip():
    """mlis_amd_encode: sampled latent = (mean + exp(0.5*clamp(logvar))*N) * scale_factor with ONE Philox call per image,
    the denoising draws continue the stream (offset 1...) like the reference's global g_rng."""
    from mlimgsynth_amd import engine
    model, side, lat = "tiny", 64, 8
    U, cond, uncond, _, _ = gen_inputs(model)
    img = np.random.default_rng(3).random((1, 3, side, side)).astype(F)
    g = engine.Generator(model, side, side, 1, n_step=10, cfg_scale=7.0, s_ancestral=1.0, f_t_ini=0.5)
    g.set_cond(cond, None, uncond, None)
    g.seed([77])
    latent = g.encode(img, sample=True)
    OP = O.Params(1234)
    V = O.vae_params(model)
    mom = O.L().orc_vae_encode_moments(OP.h, b"vae", V, O.to_ot(img))
    rnd = O.randn(77, 0, 4 * lat * lat)
    ref = O.from_ot(O.L().orc_latent_sample(mom, V, O.fptr(rnd)))
    print("encode+sample rel-L2", rel(latent, ref))
    assert rel(latent, ref) < T.EVAL
    got, _ = g.generate(None, want_images=False)            # seeds None: continue the Philox streams (offset 1)
    out, nfe = oracle_sample(model, lat, cond, uncond, None, None, "euler", 1.0, 1, 0.0, 10, 77, f_t_ini=0.5, init=ref[0], rng_offset=1)
    assert g.last_n_step() == 5 and rel(got[0], out) < T.LATENT
    g.destroy()


def test_tae_encoder_vs_oracle():
    from mlimgsynth_amd import engine
    side = 64
    img = np.random.default_rng(4).random((1, 3, side, side)).astype(F)
    g = engine.Generator("tiny", side, side, 1, n_step=2, use_tae=True)
    latent = g.encode(img, sample=False)
    OP = O.Params(1234)
    ref = O.from_ot(O.L().orc_tae_encode(OP.h, b"tae", O.to_ot(img)))
    print("tae encode rel-L2", rel(latent, ref))
    assert rel(latent, ref) < T.EVAL
    g.destroy()


def test_vae_tiling_decode_and_encode_vs_oracle():
    """MLIS_OPT_VAE_TILE (src/vae.c:245-300,333-391): overlapping tiles of tile_px + margins through tile-sized plans, interiors
    pasted in the reference's order.  256x256 image / 32x32 latent with 64-px tiles: latent tiles of 24x24 at offsets {0, 8}
    (decode), image tiles of 192x192 at offsets {0, 64} (encode).  (A dimension that one tile covers completely while the other
    is tiled keeps its last margin unwritten in the reference, src/vae.c:368-382: both dimensions are tiled here.)"""
    from mlimgsynth_amd import engine
    W, H = 256, 256
    g = engine.Generator("tiny", W, H, 1, n_step=2)
    g.set_vae_tile(64)
    rng = np.random.default_rng(12)
    z = (rng.standard_normal((1, 4, H // 8, W // 8)) * 0.5).astype(F)
    g.set_init_latent(z)
    g.decode()
    img = g.image()
    OP, V = O.Params(1234), O.vae_params("tiny")
    ref_t = O.from_ot(O.L().orc_vae_decode_tiled(OP.h, b"vae", V, O.to_ot(z), 64))
    ref_full = O.from_ot(O.L().orc_vae_decode(OP.h, b"vae", V, O.to_ot(z)))
    e_t, e_full = rel(img - 0.5, ref_t - 0.5), rel(img - 0.5, ref_full - 0.5)
    print("tiled decode vs oracle tiled", e_t, "vs oracle untiled", e_full)
    assert e_t < T.EVAL and e_full > 2 * e_t            # it really is the tiled result (tile borders differ from the full decode)
    # encode: image tiles 192x192 (64 + 2*64 margin) over the 256x256 image -> 2x2 tiles
    src = rng.random((1, 3, H, W)).astype(F)
    g.seed([5])
    lat = g.encode(src, sample=True)
    mom = O.L().orc_vae_encode_moments_tiled(OP.h, b"vae", V, O.to_ot(src), 64)
    ref = O.from_ot(O.L().orc_latent_sample(mom, V, O.fptr(O.randn(5, 0, 4 * (H // 8) * (W // 8)))))
    e = rel(lat, ref)
    print("tiled encode+sample vs oracle", e)
    assert e < T.EVAL
    g.destroy()


def test_step_counts_up_to_the_api_limit():
    """the public option STEPS accepts 0..1000 like the reference; the engine's step-sized buffers (events, scalars, noise
    draws, sigma table) follow the requested count instead of a fixed 255 cap (300 Euler-a steps, 500 requested Heun steps = 250 + 250 evaluations)."""
    from mlimgsynth_amd import engine
    import golden_cases as G
    cond, uncond, _, _ = G.gen_inputs("gen_tiny_8_20", "tiny")
    g = engine.Generator("tiny", 64, 64, 1, n_step=300, cfg_scale=7.0, s_ancestral=1.0)
    g.set_cond(cond, None, uncond, None)
    lat, _ = g.generate([3], want_images=False)
    assert np.isfinite(lat).all() and g.last_n_step() == 300 and g.last_nfe() == 600
    g.destroy()
    g = engine.Generator("tiny", 64, 64, 1, n_step=500, cfg_scale=7.0, s_ancestral=0.0, method="heun")
    g.set_cond(cond, None, uncond, None)
    lat, _ = g.generate([3], want_images=False)
    assert np.isfinite(lat).all() and g.last_n_step() == 250
    g.destroy()
    with pytest.raises(Exception):
        engine.Generator("tiny", 64, 64, 1, n_step=1001)
