"""Definition of the golden-fixture cases (tests/golden/torch_golden.npz): model configurations and the SEEDED inputs.
Shared by the generator (tools/make_torch_golden.py, which computes the expected outputs with the independent torch
restatement tools/torch_ref.py) and by the tests that check the oracle (CPU) and the HIP engine (GPU) against them.
Only data lives here: hyper-parameter tables (the public SD1.5 / SDXL / CLIP / VAE configurations, plus the shrunken
test configurations) and input generation.  Weights are the synthetic (seed 1234, name, shape) weights."""
import numpy as np

WEIGHT_SEED = 1234

UNET = {   # public configs: SD1.5 v1-inference.yaml, SDXL sd_xl_base.yaml; "tiny*": shrunken test configs of this repo
    "sd1": dict(n_ch_in=4, n_ch_out=4, n_res_blk=2, attn_res=[4, 2, 1], ch_mult=[1, 2, 4, 4], transf_depth=[1, 1, 1, 1],
                n_te=1280, n_head=8, d_head=0, n_ctx=768, n_ch=320, ch_adm_in=0),
    "sdxl": dict(n_ch_in=4, n_ch_out=4, n_res_blk=2, attn_res=[4, 2], ch_mult=[1, 2, 4], transf_depth=[1, 2, 10],
                 n_te=1280, n_head=0, d_head=64, n_ctx=2048, n_ch=320, ch_adm_in=2816),
    "tiny": dict(n_ch_in=4, n_ch_out=4, n_res_blk=1, attn_res=[2, 1], ch_mult=[1, 2], transf_depth=[1, 1],
                 n_te=256, n_head=2, d_head=0, n_ctx=64, n_ch=64, ch_adm_in=0),
    "tinyxl": dict(n_ch_in=4, n_ch_out=4, n_res_blk=2, attn_res=[2], ch_mult=[1, 2], transf_depth=[1, 2],
                   n_te=256, n_head=0, d_head=64, n_ctx=128, n_ch=64, ch_adm_in=96),
}
VAE = {
    "sd1": dict(ch_x=3, ch_z=4, ch=128, n_res=4, n_res_blk=2, ch_mult=[1, 2, 4, 4], d_embed=4, scale_factor=0.18215),
    "sdxl": dict(ch_x=3, ch_z=4, ch=128, n_res=4, n_res_blk=2, ch_mult=[1, 2, 4, 4], d_embed=4, scale_factor=0.13025),
    "tiny": dict(ch_x=3, ch_z=4, ch=64, n_res=4, n_res_blk=1, ch_mult=[1, 2, 4, 4], d_embed=4, scale_factor=0.18215),
}
CLIP = {   # OpenAI CLIP ViT-L/14 text tower, open_clip ViT-bigG-14 text tower, shrunken test tower
    "vit_l": dict(n_vocab=49408, n_token=77, d_embed=768, n_interm=3072, n_head=12, n_layer=12, tok_start=49406, tok_end=49407, tok_pad=49407),
    "vit_bigg": dict(n_vocab=49408, n_token=77, d_embed=1280, n_interm=5120, n_head=20, n_layer=32, tok_start=49406, tok_end=49407, tok_pad=0),
    "tiny": dict(n_vocab=1000, n_token=77, d_embed=64, n_interm=256, n_head=2, n_layer=3, tok_start=998, tok_end=999, tok_pad=999),
}

# (key, model, latent side, batch, sigmas)
UNET_CASES = [
    ("unet_tiny_8", "tiny", 8, 2, [14.6, 1.0]),
    ("unet_tinyxl_8", "tinyxl", 8, 2, [7.0, 0.3]),
    ("unet_sd1_16", "sd1", 16, 1, [3.0]),
    ("unet_sdxl_16", "sdxl", 16, 1, [3.0]),
]
# (key, model, latent side)
VAE_CASES = [("vae_tiny_8", "tiny", 8), ("vae_sd1_8", "sd1", 8), ("vae_sdxl_16", "sdxl", 16)]
TAE_CASES = [("tae_8", 8), ("tae_16", 16)]
# (key, clip model, prefix, clip_skip, norm, want_feat, n_tok)
CLIP_CASES = [
    ("clip_tiny", "tiny", "clip", 1, True, False, 9),
    ("clip_tiny_sdxl2", "tiny", "clip2", 2, False, True, 9),
    ("clip_vit_l", "vit_l", "clip", 1, True, False, 9),
    ("clip_vit_l_skip2", "vit_l", "clip", 2, False, False, 9),
    ("clip_bigg", "vit_bigg", "clip2", 2, False, True, 9),
]
# (key, model, latent side, steps, seed)
GEN_CASES = [("gen_tiny_8_20", "tiny", 8, 20, 42), ("gen_tinyxl_8_6", "tinyxl", 8, 6, 43)]
# (key, vae model, image side)   -- VAE encoder moments (f2)
VAE_ENC_CASES = [("vaeenc_tiny_64", "tiny", 64), ("vaeenc_sd1_64", "sd1", 64)]


# HEADLINE sizes (BASELINE.json configs[1] / [2]): the UNet at the full latent and the decoders at full resolution.  Stored in
# tests/golden/torch_golden_headline.npz; decoded images are stored REDUCED (reduce_image: 16x16 block means + 4096 sampled
# pixels) so that a 12 MB image becomes a 65 KB vector.  fp16-operand mode only.
HEADLINE_UNET_CASES = [("unet_sdxl_128", "sdxl", 128, 1, [3.0]), ("unet_sd1_64", "sd1", 64, 1, [3.0])]
HEADLINE_VAE_CASES = [("vae_sdxl_128", "sdxl", 128), ("vae_sd1_64", "sd1", 64)]
HEADLINE_TAE_CASES = [("tae_128", 128)]


def reduce_image(img, key):
    """img [1,3,H,W] -> 1-D vector: means of 16x16 blocks, then 4096 pixels at seeded positions"""
    a = np.asarray(img, np.float32)[0]
    c, h, w = a.shape
    blocks = a.reshape(c, h // 16, 16, w // 16, 16).mean(axis=(2, 4), dtype=np.float64).astype(np.float32).reshape(-1)
    r = np.random.default_rng(_seed(key) + 1)
    idx = r.integers(0, c * h * w, 4096)
    return np.concatenate([blocks, a.reshape(-1)[idx]])


def _seed(key):
    return int(np.frombuffer(key.encode().ljust(8, b"_")[:8], np.uint64)[0] % (2 ** 31))


def unet_inputs(key, model, lat, n):
    U = UNET[model]
    r = np.random.default_rng(_seed(key))
    x = (r.standard_normal((n, 4, lat, lat)) * 3).astype(np.float32)
    cond = r.standard_normal((n, 77, U["n_ctx"])).astype(np.float32)
    label = r.standard_normal((n, U["ch_adm_in"])).astype(np.float32) if U["ch_adm_in"] else None
    return x, cond, label


def vae_inputs(key, lat):
    return (np.random.default_rng(_seed(key)).standard_normal((1, 4, lat, lat)) * 0.5).astype(np.float32)


def tae_inputs(key, lat):
    return (np.random.default_rng(_seed(key)).standard_normal((1, 4, lat, lat)) * 2).astype(np.float32)


def clip_tokens(key, model, n_tok):
    K = CLIP[model]
    toks = np.random.default_rng(_seed(key)).integers(0, K["n_vocab"] - 3, n_tok).astype(np.int32)
    full = np.full(K["n_token"], K["tok_pad"], np.int32)
    full[0] = K["tok_start"]
    full[1:1 + n_tok] = toks
    full[1 + n_tok] = K["tok_end"]
    return toks, full


def gen_inputs(key, model):
    U = UNET[model]
    r = np.random.default_rng(_seed(key))
    cond = r.standard_normal((77, U["n_ctx"])).astype(np.float32)
    uncond = r.standard_normal((77, U["n_ctx"])).astype(np.float32)
    label = r.standard_normal(U["ch_adm_in"]).astype(np.float32) if U["ch_adm_in"] else None
    unlabel = r.standard_normal(U["ch_adm_in"]).astype(np.float32) if U["ch_adm_in"] else None
    return cond, uncond, label, unlabel


def image_inputs(key, side):
    return np.random.default_rng(_seed(key)).random((1, 3, side, side)).astype(np.float32)
