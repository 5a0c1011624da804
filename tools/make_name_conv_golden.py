"""Generates tests/golden/name_conv.json.gz: tensor-name conversion vectors produced by the REFERENCE's own tnconv_sd
(oracle/_ref/libtnconv_ref.so = /root/reference/src/tensor_name_conv.c compiled by oracle/Makefile).  Inputs: every
checkpoint-side tensor name of the SD1.5 / SD2 / SDXL single-file layouts (tests/ckpt_names.py applied to the engine's
parameter lists), the same with '_' and '/' separators (kohya / diffusers spellings), diffusers-style block names, and
names that must be dropped.  Run in the build container only (needs the reference build)."""
import ctypes
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ckpt_names as CN                      # noqa: E402
from loader_cases import all_internal_names  # noqa: E402

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtnconv_ref.so"))
buf = ctypes.create_string_buffer(1024)


def conv(name):
    r = ref.ref_tnconv_sd(name.encode(), buf, 1024)
    return [r, buf.value.decode()]


names = []
for model in ("sd1", "sd2", "sdxl"):
    for k in all_internal_names(model):
        e = CN.external_name(k, model)
        names.append(e[1] if isinstance(e, tuple) else e)
names = sorted(set(names))
extra = []
for n in names[::7]:
    extra.append(n.replace(".", "_"))                 # kohya-style flattening
    extra.append(n.replace(".", "/", 3))
for i in range(4):
    for j in range(3):
        for tail in ("in_layers.0.weight", "transformer_blocks.0.attn1.to_q.weight", "norm.bias", "conv.weight"):
            extra += [f"unet.down_blocks.{i}.resnets.{j}.{tail}", f"unet.down_blocks.{i}.attentions.{j}.{tail}",
                      f"unet.up_blocks.{i}.resnets.{j}.{tail}", f"unet.up_blocks.{i}.attentions.{j}.{tail}"]
    extra += [f"unet.down_blocks.{i}.downsamplers.0.conv.weight", f"unet.up_blocks.{i}.upsamplers.0.conv.weight",
              f"model.diffusion_model.down_blocks.{i}.downsamplers.0.conv.bias"]
extra += ["unet.mid_block.attentions.0.proj_in.weight", "unet.mid_block.resnets.0.in_layers.0.weight", "unet.mid_block.resnets.1.conv.bias",
          "te.text_model.encoder.layers.3.self_attn.q_proj.weight", "te1.text_model.encoder.layers.0.ln_1.weight",
          "te2.text_model.encoder.layers.11.mlp.fc1.bias", "cond_stage_model.1.transformer.text_model.final_layer_norm.weight",
          "model_ema.decay", "alphas_cumprod", "cond_stage_model.transformer.text_model.embeddings.position_ids",
          "first_stage_model.loss.logvar", "conditioner.embedders.1.model.logit_scale", "", "unet.", "model.diffusion_model.input_blocks.",
          "model.diffusion_model.input_blocks.x.0.op.weight", "lora_unet_down_blocks_0_attentions_0_proj_in.alpha"]
vec = {n: conv(n) for n in names + extra}
out = os.path.join(ROOT, "tests", "golden", "name_conv.json.gz")
with gzip.open(out, "wt", compresslevel=9) as f:
    json.dump(vec, f, sort_keys=True)
print("wrote", out, os.path.getsize(out), "bytes,", len(vec), "names;",
      sum(1 for v in vec.values() if v[0] == 0), "unused,", sum(1 for v in vec.values() if v[0] == 2), "fused in_proj")
