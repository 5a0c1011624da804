"""INDEPENDENT fp32 restatement of the hot-path graphs in plain torch (CPU), used ONLY to generate and check golden
fixtures (tests/golden/torch_golden.json).  It is written from the public model definitions the reference itself
cites —
    UNet          CompVis ldm/modules/diffusionmodules/openaimodel.py (UNetModel, ResBlock, SpatialTransformer) and
                  ldm/modules/attention.py (BasicTransformerBlock, CrossAttention, GEGLU)      <- src/unet.c:110-281
    KL-VAE        CompVis ldm/modules/diffusionmodules/model.py (Decoder/Encoder, ResnetBlock, AttnBlock)  <- src/vae.c:46-180
    TAESD         madebyollin/taesd taesd.py (Decoder, Block)                                  <- src/tae.c:24-92
    CLIP text     openai/CLIP model.py + open_clip transformer.py (text tower)                  <- src/clip.c:319-437
    sampler       k-diffusion sampling.py (sample_euler_ancestral, get_ancestral_step, to_d)   <- src/sampling.c:119-185
— with torch.nn.functional ops, NOT from oracle/*.c or the product's builders, so that a shared mis-reading in those two
(GEGLU half order, concat order, eps, clip_skip layer count, text_proj orientation, head split) shows up as a mismatch.
The only things shared with the rest of the repo are DATA: the parameter names (the reference's dotted names, which is
what a checkpoint loader would present) and the synthetic weight generator keyed by (seed, name, shape).

Numerics: fp32 throughout.  `f16_ops=True` additionally rounds the activation operand of every conv (always) and of every
linear with F16 weights to fp16 first, which is what ggml's CPU backend does (F16 im2col / mul_mat
operand conversion); it makes the comparison with the oracle tight (1e-5 class) instead of 1e-3 class.

Never imported by the product, tests import it only when torch is present AND the fixture is being (re)generated.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


class Weights:
    """name -> torch tensor (torch shape: outermost first).  `gen(name, shape, f16)` supplies the values."""

    def __init__(self, gen):
        self.gen, self.cache, self.used = gen, {}, []

    def __call__(self, name, shape, f16):
        key = (name, tuple(shape))
        if key not in self.cache:
            self.cache[key] = torch.from_numpy(np.ascontiguousarray(self.gen(name, tuple(shape), f16), dtype=np.float32))
            self.used.append((name, tuple(shape), bool(f16)))
        return self.cache[key]


class Net:
    def __init__(self, weights, f16_ops=True, linear_f16=True):
        self.W, self.f16_ops, self.linear_f16 = weights, f16_ops, linear_f16

    # ---- primitive layers (torch.nn.Linear / Conv2d / GroupNorm / LayerNorm semantics)
    def r16(self, x):
        return x.half().float() if self.f16_ops else x

    def linear(self, x, name, n_out, bias=True):
        w = self.W(name + ".weight", (n_out, x.shape[-1]), self.linear_f16)
        b = self.W(name + ".bias", (n_out,), False) if bias else None
        return F.linear(self.r16(x) if self.linear_f16 else x, w, b)

    def conv(self, x, name, ch_out, k=3, stride=1, pad=1, bias=True):
        w = self.W(name + ".weight", (ch_out, x.shape[1], k, k), True)      # conv weights are always F16 in the reference
        b = self.W(name + ".bias", (ch_out,), False) if bias else None
        return F.conv2d(self.r16(x), w, b, stride=stride, padding=pad)

    def gn(self, x, name, eps=1e-6):
        c = x.shape[1]
        return F.group_norm(x, 32, self.W(name + ".weight", (c,), False), self.W(name + ".bias", (c,), False), eps)

    def ln(self, x, name, eps=1e-5):
        d = x.shape[-1]
        return F.layer_norm(x, (d,), self.W(name + ".weight", (d,), False), self.W(name + ".bias", (d,), False), eps)

    # ---- blocks
    def resblock(self, x, emb, name, ch_out):
        """openaimodel.ResBlock (use_scale_shift_norm=False) / model.ResnetBlock (emb=None)"""
        h = self.conv(F.silu(self.gn(x, name + ".norm1")), name + ".conv1", ch_out)
        if emb is not None:
            h = h + self.linear(F.silu(emb), name + ".emb_proj", ch_out)[:, :, None, None]
        h = self.conv(F.silu(self.gn(h, name + ".norm2")), name + ".conv2", ch_out)
        if x.shape[1] != ch_out:
            x = self.conv(x, name + ".skip_conv", ch_out, k=1, pad=0)
        return x + h

    def mha(self, xq, xkv, name, d_out, d_embed, n_head, causal=False, bias=False, bias_out=True):
        """CrossAttention / nn.MultiheadAttention: softmax(q k^T / sqrt(d_head)) v per head, fp32"""
        B, Tq, _ = xq.shape
        Tk = xkv.shape[1]
        dh = d_embed // n_head
        q = self.linear(xq, name + ".q_proj", d_embed, bias).reshape(B, Tq, n_head, dh).transpose(1, 2)
        k = self.linear(xkv, name + ".k_proj", d_embed, bias).reshape(B, Tk, n_head, dh).transpose(1, 2)
        v = self.linear(xkv, name + ".v_proj", d_embed, bias).reshape(B, Tk, n_head, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        if causal:
            s = s + torch.full((Tq, Tk), float("-inf")).triu(1)
        o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Tq, d_embed)
        return self.linear(o, name + ".out_proj", d_out, bias_out)

    def basic_transformer(self, x, ctx, name, d, n_head):
        """attention.BasicTransformerBlock: self-attn, cross-attn, GEGLU feed-forward, each with a residual"""
        h = self.ln(x, name + ".norm1")
        x = self.mha(h, h, name + ".attn1", d, d, n_head) + x
        x = self.mha(self.ln(x, name + ".norm2"), ctx, name + ".attn2", d, d, n_head) + x
        h = self.linear(self.ln(x, name + ".norm3"), name + ".ff.net.0.proj", d * 4 * 2)
        val, gate = h.chunk(2, dim=-1)                                      # GEGLU: x, gate = proj(x).chunk(2); x * gelu(gate)
        h = val * F.gelu(gate, approximate="tanh")
        return self.linear(h, name + ".ff.net.2", d) + x

    def spatial_transformer(self, x, ctx, name, n_head, depth):
        """attention.SpatialTransformer (use_linear=False form: 1x1 conv projections)"""
        B, C, H, Wd = x.shape
        h = self.conv(self.gn(x, name + ".norm"), name + ".proj_in", C, k=1, pad=0)
        h = h.permute(0, 2, 3, 1).reshape(B, H * Wd, C)                     # 'b c h w -> b (h w) c'
        for i in range(depth):
            h = self.basic_transformer(h, ctx, f"{name}.transf.{i}", C, n_head)
        h = h.reshape(B, H, Wd, C).permute(0, 3, 1, 2)
        return self.conv(h, name + ".proj_out", C, k=1, pad=0) + x

    # ---- UNet (openaimodel.UNetModel.forward)
    def unet(self, P, x, t, ctx, label=None, prefix="unet"):
        p = prefix + "."
        half = P["n_ch"] // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = t[:, None].float() * freqs[None]
        temb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)        # timestep_embedding: cos | sin
        emb = self.linear(F.silu(self.linear(temb, p + "time_embed.0", P["n_te"])), p + "time_embed.2", P["n_te"])
        if P.get("ch_adm_in") and label is not None:
            emb = emb + self.linear(F.silu(self.linear(label, p + "label_embed.0", P["n_te"])), p + "label_embed.2", P["n_te"])

        def heads(ch):
            return P["n_head"] if P.get("n_head") else ch // P["d_head"]

        hs = []
        h = self.conv(x, p + "in.conv", P["n_ch"])
        hs.append(h)
        blk, ds = 0, 1
        for level, mult in enumerate(P["ch_mult"]):
            if level:
                ds *= 2
                blk += 1
                h = self.conv(h, f"{p}in.{blk}.0.conv", h.shape[1], stride=2)          # Downsample(conv)
                hs.append(h)
            for _ in range(P["n_res_blk"]):
                blk += 1
                ch = P["n_ch"] * mult
                h = self.resblock(h, emb, f"{p}in.{blk}.0", ch)
                if ds in P["attn_res"]:
                    h = self.spatial_transformer(h, ctx, f"{p}in.{blk}.1", heads(ch), P["transf_depth"][level])
                hs.append(h)
        ch = P["n_ch"] * P["ch_mult"][-1]
        top = len(P["ch_mult"]) - 1
        h = self.resblock(h, emb, p + "mid.0", ch)
        h = self.spatial_transformer(h, ctx, p + "mid.1", heads(ch), P["transf_depth"][top])
        h = self.resblock(h, emb, p + "mid.2", ch)
        ob = 0
        for level in range(top, -1, -1):
            ch = P["n_ch"] * P["ch_mult"][level]
            for j in range(P["n_res_blk"] + 1):
                h = torch.cat([h, hs.pop()], dim=1)                          # th.cat([h, hs.pop()], dim=1)
                sub = 0
                h = self.resblock(h, emb, f"{p}out.{ob}.{sub}", ch)
                sub += 1
                if ds in P["attn_res"]:
                    h = self.spatial_transformer(h, ctx, f"{p}out.{ob}.{sub}", heads(ch), P["transf_depth"][level])
                    sub += 1
                if level and j == P["n_res_blk"]:
                    h = F.interpolate(h, scale_factor=2, mode="nearest")     # Upsample
                    h = self.conv(h, f"{p}out.{ob}.{sub}.conv", ch)
                    ds //= 2
                ob += 1
        assert not hs
        return self.conv(F.silu(self.gn(h, p + "out.norm")), p + "out.conv", P["n_ch_out"])

    # ---- KL-VAE (model.Decoder / model.Encoder)
    def vae_attn(self, x, name):
        """model.AttnBlock: single head over channels"""
        B, C, H, Wd = x.shape
        h = self.gn(x, name + ".norm")
        q = self.conv(h, name + ".q", C, k=1, pad=0).reshape(B, C, H * Wd).permute(0, 2, 1)
        k = self.conv(h, name + ".k", C, k=1, pad=0).reshape(B, C, H * Wd)
        v = self.conv(h, name + ".v", C, k=1, pad=0).reshape(B, C, H * Wd)
        w = torch.softmax((q @ k) * (C ** -0.5), dim=2)                      # [b, hw_q, hw_k]
        h = (v @ w.permute(0, 2, 1)).reshape(B, C, H, Wd)
        return x + self.conv(h, name + ".proj_out", C, k=1, pad=0)

    def vae_decode(self, V, z, prefix="vae"):
        p = prefix + "."
        h = self.conv(z / V["scale_factor"], p + "post_quant_conv", V["d_embed"], k=1, pad=0)
        d = p + "decoder."
        ch = V["ch"] * V["ch_mult"][-1]
        h = self.conv(h, d + "conv_in", ch)
        h = self.resblock(h, None, d + "mid.block_1", ch)
        h = self.vae_attn(h, d + "mid.attn_1")
        h = self.resblock(h, None, d + "mid.block_2", ch)
        for i in range(V["n_res"] - 1, -1, -1):
            ch = V["ch"] * V["ch_mult"][i]
            for j in range(V["n_res_blk"] + 1):
                h = self.resblock(h, None, f"{d}up.{i}.block.{j}", ch)
            if i:
                h = self.conv(F.interpolate(h, scale_factor=2, mode="nearest"), f"{d}up.{i}.upsample.conv", ch)
        h = self.conv(F.silu(self.gn(h, d + "norm_out")), d + "conv_out", V["ch_x"])
        return (h + 1) / 2

    def vae_encode_moments(self, V, img, prefix="vae"):
        """model.Encoder + quant_conv: image in [0,1] -> moments [2*ch_z] (mean | logvar).  Downsample pads (0,1,0,1)."""
        p = prefix + "."
        e = p + "encoder."
        h = self.conv(img * 2 - 1, e + "conv_in", V["ch"])
        ch = V["ch"]
        for i in range(V["n_res"]):
            ch = V["ch"] * V["ch_mult"][i]
            for j in range(V["n_res_blk"]):
                h = self.resblock(h, None, f"{e}down.{i}.block.{j}", ch)
            if i + 1 != V["n_res"]:
                h = self.conv(F.pad(h, (0, 1, 0, 1)), f"{e}down.{i}.downsample.conv", ch, stride=2, pad=0)
        h = self.resblock(h, None, e + "mid.block_1", ch)
        h = self.vae_attn(h, e + "mid.attn_1")
        h = self.resblock(h, None, e + "mid.block_2", ch)
        h = self.conv(F.silu(self.gn(h, e + "norm_out")), e + "conv_out", V["ch_z"] * 2)
        return self.conv(h, p + "quant_conv", V["ch_z"] * 2, k=1, pad=0)

    # ---- TAESD decoder (taesd.Decoder)
    def tae_block(self, x, name):
        h = F.relu(self.conv(x, name + ".conv.0", 64))
        h = F.relu(self.conv(h, name + ".conv.2", 64))
        return F.relu(self.conv(h, name + ".conv.4", 64) + x)

    def tae_decode(self, z, prefix="tae"):
        L = prefix + ".decoder.layers."
        h = torch.tanh(z / 3) * 3
        i = 0
        h = F.relu(self.conv(h, f"{L}{i}", 64)); i += 2
        for _ in range(3):
            for _ in range(3):
                h = self.tae_block(h, f"{L}{i}"); i += 1
            h = F.interpolate(h, scale_factor=2, mode="nearest"); i += 1
            h = self.conv(h, f"{L}{i}", 64, bias=False); i += 1
        h = self.tae_block(h, f"{L}{i}"); i += 1
        return self.conv(h, f"{L}{i}", 3)

    # ---- CLIP text tower (CLIP.encode_text without the final projection; open_clip TextTransformer)
    def clip_text(self, K, tokens, prefix, clip_skip=1, norm=True):
        p = prefix + ".text."
        d = K["d_embed"]
        tw = self.W(p + "embed.token.weight", (K["n_vocab"], d), self.linear_f16)
        pw = self.W(p + "embed.position.weight", (K["n_token"], d), False)
        x = tw[tokens.long()] + pw[None]
        n_layer = K["n_layer"] - (clip_skip - 1 if clip_skip > 1 else 0)
        act = (lambda v: F.gelu(v, approximate="tanh")) if d in (1024, 1280) else (lambda v: v * torch.sigmoid(1.702 * v))
        for i in range(n_layer):
            l = f"{p}encoder.layers.{i}"
            h = self.ln(x, l + ".norm1")
            x = x + self.mha(h, h, l + ".attn", d, d, K["n_head"], causal=True, bias=True)
            h = self.linear(act(self.linear(self.ln(x, l + ".norm2"), l + ".mlp.fc1", K["n_interm"])), l + ".mlp.fc2", d)
            x = x + h
        if norm:
            x = self.ln(x, p + "ln_final")
        return x

    def clip_feat(self, K, tokens, prefix, i_tok_end):
        """pooled feature: all layers + ln_final, row of the end token, x @ text_projection (open_clip)"""
        x = self.clip_text(K, tokens, prefix, clip_skip=1, norm=True)
        proj = self.W(prefix + ".text.text_proj", (K["d_embed"], K["d_embed"]), False)
        return x[:, i_tok_end] @ proj


def sdxl_label(feat, width, height):
    """SDXL vector conditioning (sgm GeneralConditioner): pooled | emb(orig h, w) | emb(crop 0, 0) | emb(target h, w)"""
    def emb(v):
        half = 128
        f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        a = torch.tensor([float(v)]) * f
        return torch.cat([torch.cos(a), torch.sin(a)])
    parts = [feat.reshape(-1)] + [emb(v) for v in (height, width, 0, 0, height, width)]
    return torch.cat(parts)


def euler_ancestral(denoise_eps, x0_noise, sigmas, noise_fn, eta=1.0):
    """k-diffusion sample_euler_ancestral in eps form.  denoise_eps(x, sigma) -> d = (x - denoised)/sigma = eps.
    The reference starts from zeros + noise*sigma_0 (no sqrt(1+sigma^2) scaling, src/mlimgsynth.c:1669)."""
    x = x0_noise * sigmas[0]
    n = len(sigmas) - 1
    for i in range(n):
        s1, s2 = np.float32(sigmas[i]), np.float32(sigmas[i + 1])
        d = denoise_eps(x, float(s1))
        s_up = np.float32(min(float(s2), eta * math.sqrt(float(s2) ** 2 * (float(s1) ** 2 - float(s2) ** 2) / float(s1) ** 2)))
        s_down = np.float32(math.sqrt(float(s2) ** 2 - float(s_up) ** 2))
        x = x + d * float(s_down - s1)
        if s_up > 0 and i + 1 != n:
            x = x + noise_fn(i + 1) * float(s_up)
    return x
