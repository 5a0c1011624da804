"""Checkpoint-side tensor names (test data generator): maps the engine's internal dotted parameter names to the names the
public checkpoints use — CompVis / sgm "original" layout (SD1.x, SD2.x, SDXL single-file .safetensors): LDM UNet
(model.diffusion_model.*), first_stage_model.*, HF-transformers CLIP text tower (cond_stage_model.transformer.* /
conditioner.embedders.0.transformer.*), open_clip text tower (cond_stage_model.model.* / conditioner.embedders.1.model.*
with the fused attn.in_proj tensors).  Written from the layouts of those public files, independent of
csrc/host/name_conv.c; the forward direction is what tests/test_loader_*.py check against the reference build."""
import re


def unet_ext(name):
    """'unet.<...>' -> 'model.diffusion_model.<...>'"""
    s = name[len("unet."):]
    s = re.sub(r"^label_embed\.", "label_emb.0.", s)
    s = re.sub(r"^in\.conv\.", "input_blocks.0.0.", s)
    s = re.sub(r"^out\.norm\.", "out.0.", s)
    s = re.sub(r"^out\.conv\.", "out.2.", s)
    m = re.match(r"^(in|out|mid)\.(\d+)\.(?:(\d+)\.)?(.*)$", s)
    if m and not s.startswith(("out.0.", "out.2.")) or (m and name.startswith(("unet.out.0.", "unet.out.2."))):
        kind, a, b, rest = m.groups()
        if kind == "mid":
            head = f"middle_block.{a}."
            rest = (b + "." if b is not None else "") + rest
        else:
            head = ("input_blocks." if kind == "in" else "output_blocks.") + f"{a}.{b}."
        rest = re.sub(r"^norm1\.", "in_layers.0.", rest)
        rest = re.sub(r"^conv1\.", "in_layers.2.", rest)
        rest = re.sub(r"^norm2\.", "out_layers.0.", rest)
        rest = re.sub(r"^conv2\.", "out_layers.3.", rest)
        rest = re.sub(r"^emb_proj\.", "emb_layers.1.", rest)
        rest = re.sub(r"^skip_conv\.", "skip_connection.", rest)
        if kind == "in":
            rest = re.sub(r"^conv\.", "op.", rest)          # Downsample.op; Upsample keeps .conv
        rest = re.sub(r"^transf\.", "transformer_blocks.", rest)
        rest = rest.replace(".q_proj.", ".to_q.").replace(".k_proj.", ".to_k.").replace(".v_proj.", ".to_v.").replace(".out_proj.", ".to_out.0.")
        s = head + rest
    return "model.diffusion_model." + s


def vae_ext(name):
    s = name[len("vae."):]
    s = re.sub(r"^((?:en|de)coder\.(?:up|down)\.\d+\.block\.\d+\.)skip_conv\.", r"\1nin_shortcut.", s)
    return "first_stage_model." + s


def clip_hf_ext(name, root):
    """'<clip>.text.<...>' -> root + 'transformer.text_model.<...>' (HF CLIPTextModel)"""
    s = name.split(".text.", 1)[1]
    s = s.replace("embed.token.", "embeddings.token_embedding.").replace("embed.position.", "embeddings.position_embedding.")
    s = s.replace(".norm1.", ".layer_norm1.").replace(".norm2.", ".layer_norm2.").replace(".attn.", ".self_attn.")
    s = s.replace("ln_final.", "final_layer_norm.")
    s = s.replace("text_proj", "text_projection")
    return root + "transformer.text_model." + s


def clip_openclip_ext(name, root):
    """'<clip>.text.<...>' -> root + 'model.<...>' (open_clip); q/k/v_proj are returned as ('FUSE', fused_name, part)"""
    s = name.split(".text.", 1)[1]
    if s == "embed.token.weight":
        return root + "model.token_embedding.weight"
    if s == "embed.position.weight":
        return root + "model.positional_embedding"
    if s == "text_proj":
        return root + "model.text_projection"
    if s.startswith("ln_final."):
        return root + "model." + s
    m = re.match(r"^encoder\.layers\.(\d+)\.(.*)$", s)
    i, rest = m.groups()
    base = f"{root}model.transformer.resblocks.{i}."
    mm = re.match(r"^attn\.([qkv])_proj\.(weight|bias)$", rest)
    if mm:
        return ("FUSE", base + "attn.in_proj_" + mm.group(2), "qkv".index(mm.group(1)))
    rest = rest.replace("norm1.", "ln_1.").replace("norm2.", "ln_2.").replace("mlp.fc1.", "mlp.c_fc.").replace("mlp.fc2.", "mlp.c_proj.")
    return base + rest


def external_name(name, model):
    """internal name -> checkpoint name for `model` in ('sd1', 'sd2', 'sdxl')"""
    if name.startswith("unet."):
        return unet_ext(name)
    if name.startswith("vae."):
        return vae_ext(name)
    if name.startswith("clip."):
        if model == "sd1":
            return clip_hf_ext(name, "cond_stage_model.")
        if model == "sd2":
            return clip_openclip_ext(name, "cond_stage_model.")
        return clip_hf_ext(name, "conditioner.embedders.0.")
    if name.startswith("clip2."):
        return clip_openclip_ext(name, "conditioner.embedders.1.")
    raise ValueError(name)
