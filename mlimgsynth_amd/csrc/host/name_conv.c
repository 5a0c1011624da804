/* Checkpoint tensor names -> the engine's dotted parameter names.  Behavioural re-creation of the reference's tnconv_sd
 * (src/tensor_name_conv.c:274-324 and the sub-grammars it dispatches to, :84-272): CompVis / SDXL (sgm) / diffusers UNet
 * naming, HF-transformers and open_clip text-tower naming, CompVis VAE naming.  Pinned against the reference's own
 * tensor_name_conv.c (built by oracle/Makefile into oracle/_ref) over every tensor name of the SD1.5 / SD2 / SDXL layouts:
 * tests/test_loader_cpu.py, fixture tests/golden/name_conv.json.
 *
 * Implemented as a rule interpreter: a rule is (pattern, replacement, action) where '.' in a pattern also matches '_' and
 * '/' (diffusers / kohya spellings), "#" matches a number followed by a separator.  Rule sets are tried top to bottom; the
 * unmatched tail of the name is copied.  Return: 0 unused, 1 converted, 2 converted and it is an open_clip fused
 * attention in_proj tensor that the caller must split in three (TNCONV_R_QKV_PROJ).
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"

typedef struct { const char* s; size_t n; char* o; size_t on, cap; int ovf; } Cur;

static int is_sep(char c) { return c == '.' || c == '_' || c == '/'; }

static void emit(Cur* c, const char* s, size_t n)
{
	if (c->on + n + 1 > c->cap) { c->ovf = 1; return; }
	memcpy(c->o + c->on, s, n); c->on += n; c->o[c->on] = 0;
}
static void emitz(Cur* c, const char* s) { emit(c, s, strlen(s)); }

/* prefix match where '.' in the pattern matches any separator; consumes on success */
static int take(Cur* c, const char* pat)
{
	const size_t l = strlen(pat);
	if (c->n < l) return 0;
	for (size_t i=0;i<l;++i) if (!(pat[i] == c->s[i] || (pat[i] == '.' && is_sep(c->s[i])))) return 0;
	c->s += l; c->n -= l;
	return 1;
}
static int peek(const Cur* c, const char* pat) { Cur t = *c; return take(&t, pat); }
static int rep(Cur* c, const char* pat, const char* with) { if (!take(c, pat)) return 0; emitz(c, with); return 1; }
static int keep(Cur* c, const char* pat) { return rep(c, pat, pat); }

/* digits followed by a separator; `min_digits` 1: the reference's push variant needs at least one digit, the get variant
 * accepts an empty number (then atoi gives 0) */
static int number(Cur* c, int* val, int push, int min_digits)
{
	size_t i = 0;
	while (i < c->n && c->s[i] >= '0' && c->s[i] <= '9') i++;
	if (i == c->n || !is_sep(c->s[i]) || (int)i < min_digits) return 0;
	if (val) *val = atoi(c->s);
	if (push) { emit(c, c->s, i); emitz(c, "."); }
	c->s += i + 1; c->n -= i + 1;
	return 1;
}

static int finish(Cur* c, int r) { if (r > 0) emit(c, c->s, c->n); return r; }

/* ---- text towers */
static int clip_hf(Cur* c)          /* HF transformers CLIPTextModel (SD1 cond_stage_model, SDXL embedders.0) */
{
	int r = 0;
	if (rep(c, "transformer.text_model.", "text.")) {
		if (rep(c, "embeddings.", "embed.")) {
			if (rep(c, "position_embedding.", "position.") || rep(c, "token_embedding.", "token.")) r = 1;
		} else if (keep(c, "encoder.layers.")) {
			number(c, NULL, 1, 1);
			if (rep(c, "layer_norm1.", "norm1.") || rep(c, "layer_norm2.", "norm2.") || rep(c, "self_attn.", "attn.") || keep(c, "mlp.")) r = 1;
		} else if (rep(c, "final_layer_norm.", "ln_final.") || rep(c, "text_projection", "text_proj")) r = 1;
	}
	return finish(c, r);
}

static int clip_openclip(Cur* c)    /* open_clip text tower (SD2 cond_stage_model.model, SDXL embedders.1.model) */
{
	int r = 0;
	if (rep(c, "model.", "text.")) {
		if (keep(c, "ln_final.") || rep(c, "token_embedding.", "embed.token.") || rep(c, "positional_embedding", "embed.position.weight") ||
		    rep(c, "text_projection", "text_proj")) r = 1;
		else if (rep(c, "transformer.resblocks.", "encoder.layers.")) {
			number(c, NULL, 1, 1);
			if (rep(c, "ln_1.", "norm1.") || rep(c, "ln_2.", "norm2.")) r = 1;
			else if (keep(c, "attn.")) {
				if (keep(c, "in_proj_bias") || keep(c, "in_proj_weight")) r = TNCONV_R_QKV_PROJ;
				else if (keep(c, "out_proj.")) r = 1;
			}
			else if (rep(c, "mlp.c_fc.", "mlp.fc1.") || rep(c, "mlp.c_proj.", "mlp.fc2.")) r = 1;
		}
	}
	return finish(c, r);
}

static int clip_kohya(Cur* c)       /* "te." / "te1." / "te2." spelling (diffusers-converted) */
{
	int r = 0;
	if (rep(c, "text_model.", "text.") && keep(c, "encoder.layers.")) {
		number(c, NULL, 1, 1);
		if (rep(c, "ln_1.", "norm1.") || rep(c, "ln_2.", "norm2.") || rep(c, "self_attn.", "attn.") || keep(c, "mlp.")) r = 1;
	}
	return finish(c, r);
}

/* ---- VAE (CompVis first_stage_model) */
static int vae(Cur* c)
{
	int r = 0;
	if (keep(c, "decoder.")) {
		r = 1;
		if (keep(c, "up.") && number(c, NULL, 1, 1) && keep(c, "block.") && number(c, NULL, 1, 1)) rep(c, "nin_shortcut.", "skip_conv.");
	} else if (keep(c, "encoder.")) {
		r = 1;
		if (keep(c, "down.") && number(c, NULL, 1, 1) && keep(c, "block.") && number(c, NULL, 1, 1)) rep(c, "nin_shortcut.", "skip_conv.");
	} else if (keep(c, "quant_conv.") || keep(c, "post_quant_conv.")) r = 1;
	return finish(c, r);
}

/* ---- UNet: the part after "<in|out>.<i>.<j>." / "mid.<j>." */
static int unet_block(Cur* c)
{
	static const char* const pairs[][2] = {
		{"in_layers.0.", "norm1."}, {"in_layers.2.", "conv1."}, {"out_layers.0.", "norm2."}, {"out_layers.3.", "conv2."},
		{"emb_layers.1.", "emb_proj."}, {"skip_connection.", "skip_conv."}, {"op.", "conv."}, {"norm.", "norm."},
		{"proj_in.", "proj_in."}, {"proj_out.", "proj_out."}, {"conv.", "conv."},
	};
	int r = 0;
	if (rep(c, "transformer_blocks.", "transf.")) {
		number(c, NULL, 1, 1);
		if (keep(c, "attn1.") || keep(c, "attn2.")) {
			if (!rep(c, "to_q.", "q_proj.") && !rep(c, "to_k.", "k_proj.") && !rep(c, "to_v.", "v_proj.")) rep(c, "to_out.0.", "out_proj.");
			r = 1;
		} else if (keep(c, "ff.")) {
			if (keep(c, "net.0.") || keep(c, "net.2.")) r = 1;
		} else if (keep(c, "norm1.") || keep(c, "norm2.") || keep(c, "norm3.")) r = 1;
	} else {
		for (size_t i=0; i<sizeof(pairs)/sizeof(pairs[0]) && !r; ++i) if (rep(c, pairs[i][0], pairs[i][1])) r = 1;
	}
	return finish(c, r);
}

static void emit_fmt2(Cur* c, int a, int b) { char t[48]; snprintf(t, sizeof(t), "%d.%d.", a, b); emitz(c, t); }

static int unet(Cur* c)
{
	int r = 0, n1 = 0, n2 = 0, n3 = 0;
	if (keep(c, "time_embed.") || rep(c, "label_emb.0.", "label_embed.") || rep(c, "input_blocks.0.0.", "in.conv.") ||
	    rep(c, "out.0.", "out.norm.") || rep(c, "out.2.", "out.conv.")) r = 1;
	else if ((rep(c, "input_blocks.", "in.") && number(c, NULL, 1, 1)) || (rep(c, "output_blocks.", "out.") && number(c, NULL, 1, 1)) ||
	         rep(c, "middle_block.", "mid.")) {
		number(c, NULL, 1, 1);
		return unet_block(c);
	}
	/* diffusers layout (after diffusers/scripts/convert_diffusers_to_original_stable_diffusion.py) */
	else if (rep(c, "down_blocks.", "in.")) {
		if (!number(c, &n1, 0, 0)) return 0;
		if (take(c, "downsamplers.0.conv.")) { char t[48]; snprintf(t, sizeof(t), "%d.0.op.", 3*(n1+1)); emitz(c, t); }
		else {
			if (take(c, "attentions.")) n2 = 1; else if (take(c, "resnets.")) n2 = 0; else return 0;
			if (!number(c, &n3, 0, 0)) return 0;
			emit_fmt2(c, 3*n1 + n3 + 1, n2);
		}
		return unet_block(c);
	}
	else if (rep(c, "up_blocks.", "out.")) {
		if (!number(c, &n1, 0, 0)) return 0;
		if (take(c, "upsamplers.0.")) emit_fmt2(c, 3*n1 + 2, n1 == 0 ? 1 : 2);
		else {
			if (take(c, "attentions.")) n2 = 1; else if (take(c, "resnets.")) n2 = 0; else return 0;
			if (!number(c, &n3, 0, 0)) return 0;
			emit_fmt2(c, 3*n1 + n3, n2);
		}
		return unet_block(c);
	}
	else if (rep(c, "mid_block.", "mid.")) {
		if (rep(c, "attentions.0.", "1.")) return unet_block(c);
		if (rep(c, "resnets.0.", "0.") || rep(c, "resnets.1.", "2.")) r = 1;
	}
	return finish(c, r);
}

MLB_API int tnconv_sd(const char* name, char* out, size_t out_size)
{
	if (!name || !out || !out_size) return -1;
	Cur c = { name, strlen(name), out, 0, out_size, 0 };
	out[0] = 0;
	int r = 0;
	if (rep(&c, "cond_stage_model.1.", "clip2.")) r = clip_hf(&c);                       /* sd.cpp-style SDXL second tower */
	else if (rep(&c, "cond_stage_model.", "clip.")) {
		if (peek(&c, "transformer.text_model.")) r = clip_hf(&c);                        /* SD1 */
		else if (peek(&c, "model.")) r = clip_openclip(&c);                              /* SD2 */
	}
	else if (rep(&c, "te.", "clip.")) r = clip_kohya(&c);
	else if (rep(&c, "conditioner.embedders.0.", "clip.")) r = clip_hf(&c);              /* SDXL */
	else if (rep(&c, "conditioner.embedders.1.", "clip2.")) r = clip_openclip(&c);
	else if (rep(&c, "te1.", "clip.")) r = clip_kohya(&c);
	else if (rep(&c, "te2.", "clip2.")) r = clip_kohya(&c);
	else if (rep(&c, "first_stage_model.", "vae.")) r = vae(&c);
	else if (rep(&c, "model.diffusion_model.", "unet.") || keep(&c, "unet.")) r = unet(&c);
	if (c.ovf) return mlsd_set_error(-1, "tensor name too long: %s", name);
	return r;
}
