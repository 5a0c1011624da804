/* CLIP text encoder on the MI355X plan builder — re-creation of the graph part of the
 * reference's src/clip.c:23-57,319-488 (the BPE tokenizer, src/clip.c:59-315, is host-side
 * integer code outside this file).  Batched over prompts; the graph and its weights can be kept
 * resident (ClipEncoder) instead of being rebuilt per prompt as in clip_text_encode().
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>

#define YES true
#define MLN(NAME,X)  mlctx_tensor_add(C, (NAME), (X))

MLB_API int clip_params_get(const char* model, ClipParams* K)
{	/* src/clip.c:23-57 */
	memset(K, 0, sizeof(*K));
	K->n_vocab=49408; K->n_token=77; K->tok_start=49406; K->tok_end=49407;
	if (!strcmp(model,"vit_l"))         { K->d_embed=768;  K->n_interm=3072; K->n_head=12; K->n_layer=12; K->tok_pad=49407; }
	else if (!strcmp(model,"vit_h"))    { K->d_embed=1024; K->n_interm=4096; K->n_head=16; K->n_layer=24; K->tok_pad=0; }
	else if (!strcmp(model,"vit_bigg")) { K->d_embed=1280; K->n_interm=5120; K->n_head=20; K->n_layer=32; K->tok_pad=0; }
	else if (!strcmp(model,"tiny"))     { K->n_vocab=1000; K->d_embed=64; K->n_interm=256; K->n_head=2; K->n_layer=3;
	                                      K->tok_start=998; K->tok_end=999; K->tok_pad=999; }
	else return mlsd_set_error(-1, "unknown CLIP model '%s'", model);
	return 1;
}

/* mlb_clip_embeddings, src/clip.c:319-344 */
static MLTensor* mlb_clip_embeddings(MLCtx* C, MLTensor* tokens, int d_embed, int n_vocab, int n_token)
{
	mlctx_block_begin(C);
	const int n = (int)(tokens->ne[1] > 0 ? tokens->ne[1] : 1) * (int)(tokens->ne[2] > 0 ? tokens->ne[2] : 1);
	MLParam *tw = mlctx_param_new(C, "token.weight", MLT_F16, d_embed, n_vocab, 1, 1, 0, 0, 0);
	const void *twd = tw->dev;
	MLParam *pw = mlctx_param_new(C, "position.weight", MLT_F32, d_embed, n_token, 1, 1, 0, 0, 0);
	const float *pwd = (const float*)pw->dev;
	MLTensor *x = mlt_new(C, n, 1, n_token, d_embed);
	x->sz32 = (size_t)n * n_token * d_embed * 4; x->d32 = (float*)mlctx_dalloc(C, x->sz32, 0); x->ld32 = d_embed;
	MLOp *op = mlctx_op_new(C, OP_CLIP_EMBED, "clip_embed");
	op->u.cemb.tok = (const int32_t*)tokens->in_stage; op->u.cemb.n = n; op->u.cemb.T = n_token; op->u.cemb.d = d_embed;
	op->u.cemb.tw = twd; op->u.cemb.pw = pwd; op->u.cemb.out = x->d32;
	return x;
}

/* mlb_clip_mlp, src/clip.c:346-360 */
static MLTensor* mlb_clip_mlp(MLCtx* C, MLTensor* x, int d_model, int n_interm, MLTensor* resid)
{
	mlctx_block_begin(C);
	MLEpilogue act = {0};
	act.act = (d_model == 1024 || d_model == 1280) ? MLSD_ACT_GELU : MLSD_ACT_GELU_QUICK;   /* SD2/SDXL vs SD1 */
	MLTensor *h = MLN("fc1", mlb_linear_ex(C, x, n_interm, YES, &act, 0));
	MLEpilogue ep = {0}; ep.resid = resid;
	MLTensor *y = MLN("fc2", mlb_linear_ex(C, h, d_model, YES, &ep, 0));
	mlb_release(C, h);
	return y;
}

/* mlb_clip_layer, src/clip.c:362-377; consumes x */
static MLTensor* mlb_clip_layer(MLCtx* C, MLTensor* x, int d_model, int n_head, int n_interm, bool mask)
{
	mlctx_block_begin(C);
	MLTensor *n = MLN("norm1", mlb_layer_norm_ex(C, x, 0, 0));
	MLTensor *a = MLN("attn", mlb_attn_mhead_ex(C, n, n, n, d_model, d_model, n_head, mask, YES, YES, x));
	mlb_release(C, n);
	if (!a || !mlt_need32(C, a)) return NULL;
	mlb_release(C, x);
	n = MLN("norm2", mlb_layer_norm_ex(C, a, 0, 0));
	MLTensor *y = MLN("mlp", mlb_clip_mlp(C, n, d_model, n_interm, a));
	mlb_release(C, n);
	if (!y || !mlt_need32(C, y)) return NULL;
	mlb_release(C, a);
	return y;
}

/* mlb_clip_text, src/clip.c:395-416 */
static MLTensor* mlb_clip_text(MLCtx* C, MLTensor* tokens, const ClipParams* P, int clip_skip, bool norm, int tap_skip, void** tap_dev)
{
	char name[64];
	mlctx_block_begin(C);
	MLTensor *x = MLN("embed", mlb_clip_embeddings(C, tokens, P->d_embed, P->n_vocab, P->n_token));
	int n_layer = P->n_layer;
	if (clip_skip > 1) n_layer -= clip_skip - 1;
	const int tap_after = (tap_dev && tap_skip > 0) ? P->n_layer - (tap_skip - 1) : -1;   /* layers run before the tap */
	mlctx_block_begin(C);   /* mlb_clip_encoder :380-393 */
	for (int i=0; i<n_layer; ++i) {
		sprintf(name, "layers.%d", i);
		x = MLN(name, mlb_clip_layer(C, x, P->d_embed, P->n_head, P->n_interm, true));
		if (!x) return NULL;
		if (i + 1 == tap_after) {   /* the next layer consumes x: keep a copy (77 x d floats per prompt) */
			const size_t nb = (size_t)x->n * x->h * x->w * x->c * sizeof(float);
			if (!mlt_need32(C, x) || x->ld32 != x->c) { mlctx_fail(C, "clip tap: unexpected layout"); return NULL; }
			*tap_dev = mlctx_dalloc(C, nb, 1);
			MLOp *op = mlctx_op_new(C, OP_COPY_F32, "clip_tap_copy");
			op->u.copy.src = x->d32; op->u.copy.dst = *tap_dev; op->u.copy.nbytes = nb;
		}
	}
	MLN("encoder", x);
	if (norm) {
		MLTensor *y = MLN("ln_final", mlb_layer_norm_ex(C, x, 0, 1));
		mlb_release(C, x);
		x = y;
	}
	return x;
}

static int param_to_host(MLCtx* C, const char* key, float* out, size_t n)
{
	for (int i=0;i<C->n_params;++i) if (C->params[i].key && !strcmp(C->params[i].key, key)) {
		MLParam *p = &C->params[i];
		if (p->type != MLT_F32 || p->layout != 0 || p->dev_elems != n) return mlctx_fail(C, "param_to_host(%s): unsupported", key);
		if (C->pstream && (const char*)p->dev >= MLW_VBASE && (const char*)p->dev < MLW_VBASE + C->pv_size)      /* a virtual (streamed) address is never dereferenced */
			return mlctx_fail(C, "param_to_host(%s): the parameter is streamed (mlctx_set_weight_streaming): not supported by this builder", key);
		if (mlsd_memcpy(out, p->dev, n*4, 1, C->stream) || mlsd_stream_sync(C->stream)) return -1;
		return 1;
	}
	return mlctx_fail(C, "unknown parameter '%s'", key);
}

/* ------------------------------------------------------------------ resident encoder */
MLB_API int clip_encoder_init_ex(ClipEncoder* E, MLCtx* C, const ClipParams* P, const char* tprefix, unsigned n_prompt,
	int clip_skip, bool norm, bool want_feat, int tap_skip)
{
	memset(E, 0, sizeof(*E));
	if (want_feat) { clip_skip = -1; norm = true; }          /* src/clip.c:446 */
	if (tap_skip > 0 && (clip_skip > 1 || tap_skip > P->n_layer)) return mlsd_set_error(-1, "clip_encoder_init_ex: a tap needs the whole stack");
	E->C = C; E->P = *P; E->n_prompt = n_prompt; E->want_feat = want_feat;
	snprintf(E->prefix, sizeof(E->prefix), "%s", tprefix);
	mlctx_begin(C, "CLIP text encode");
	mlctx_set_tprefix(C, tprefix);
	E->t_tokens = mlctx_input_new_seq(C, "tokens", MLT_I32, P->n_token, n_prompt, 1);
	E->t_embed = mlb_clip_text(C, E->t_tokens, P, clip_skip, norm, tap_skip, tap_skip > 0 ? &E->tap_dev : NULL);
	if (!E->t_embed) return -1;
	if (want_feat) mlctx_param_new(C, "text_proj", MLT_F32, P->d_embed, P->d_embed, 1, 1, 0, 0, 0);   /* mlb_clip_text_proj :418-427 */
	mlctx_tensor_add(C, "text", E->t_embed);
	return mlctx_prep(C);
}

MLB_API int clip_encoder_init(ClipEncoder* E, MLCtx* C, const ClipParams* P, const char* tprefix, unsigned n_prompt,
	int clip_skip, bool norm, bool want_feat)
{
	return clip_encoder_init_ex(E, C, P, tprefix, n_prompt, clip_skip, norm, want_feat, 0);
}

MLB_API void clip_encoder_free(ClipEncoder* E) { if (E) { free(E->text_proj_host); E->text_proj_host = NULL; } }

MLB_API int clip_encoder_run_ex(ClipEncoder* E, unsigned n_used, const int* n_tok, const int32_t* const* toks, float* embed, float* tap, float* feat)
{
	MLCtx *C = E->C;
	const ClipParams *P = &E->P;
	const int NT = P->n_token, d = P->d_embed;
	const unsigned n_prompt = E->n_prompt;
	if (n_used < 1 || n_used > n_prompt) return mlsd_set_error(-1, "clip_encoder_run_ex: %u prompts, encoder built for %u", n_used, n_prompt);
	for (unsigned p=0;p<n_used;++p) if (n_tok[p] < 0 || n_tok[p] + 2 > NT) return mlsd_set_error(-1, "prompt too long (max: %d)", NT - 2);   /* :449-450 */
	if (feat && !E->want_feat) return mlsd_set_error(-1, "clip_encoder_run: encoder built without the pooled feature");
	if (tap && !E->tap_dev) return mlsd_set_error(-1, "clip_encoder_run: encoder built without a tap");
	int32_t *tokens = (int32_t*)malloc(sizeof(int32_t) * NT * n_prompt);
	for (unsigned p=0; p<n_prompt; ++p) {                    /* :451-455; slots past n_used repeat prompt 0 (their output is unused) */
		const unsigned q = p < n_used ? p : 0;
		int32_t *t = tokens + (size_t)p*NT;
		t[0] = P->tok_start;
		if (n_tok[q]) memcpy(t+1, toks[q], sizeof(int32_t)*n_tok[q]);
		t[n_tok[q]+1] = P->tok_end;
		for (int i=n_tok[q]+2; i<NT; ++i) t[i] = P->tok_pad;
	}
	int R = 1;
	float *tmp = NULL;
	const size_t ne = (size_t)n_used * NT * d;
	if (mlctx_input_set(C, E->t_tokens, tokens, sizeof(int32_t)*NT*n_prompt) < 0) { R = -1; goto end; }
	if (mlctx_compute_checked(C) < 0) { R = -1; goto end; }
	tmp = (float*)malloc((size_t)n_prompt * NT * d * 4);
	if (mlctx_output_get(C, E->t_embed, tmp, (size_t)n_prompt*NT*d*4) < 0) { R = -1; goto end; }
	if (embed) memcpy(embed, tmp, ne*4);
	if (tap && (mlsd_memcpy(tap, E->tap_dev, ne*4, 1, C->stream) || mlsd_stream_sync(C->stream))) { R = -1; goto end; }
	if (feat) {
		/* feat = text_proj^T . x[EOS]  (:428-434): F32 weights and activations in the reference; done on the
		 * host in double (d*d MACs per prompt) */
		if (!E->text_proj_host) {
			char key[96]; snprintf(key, sizeof(key), "%s.text.text_proj", E->prefix);
			E->text_proj_host = (float*)malloc((size_t)d*d*4);
			if (param_to_host(C, key, E->text_proj_host, (size_t)d*d) < 0) { free(E->text_proj_host); E->text_proj_host = NULL; R = -1; goto end; }
		}
		const float *tp = E->text_proj_host;
		for (unsigned p=0; p<n_used; ++p) {
			const float *xe = tmp + ((size_t)p*NT + n_tok[p] + 1) * d;
			for (int j=0;j<d;++j) { double s=0; for (int i=0;i<d;++i) s += (double)tp[j + (size_t)d*i] * xe[i]; feat[(size_t)p*d + j] = (float)s; }
		}
	}
end:
	free(tokens); free(tmp);
	return R;
}

MLB_API int clip_encoder_run(ClipEncoder* E, unsigned n_tok, const int32_t* toks, float* embed, float* feat)
{	/* all n_prompt prompts, same length */
	const int32_t *tp[16]; int nt[16];
	if (E->n_prompt > 16) return mlsd_set_error(-1, "clip_encoder_run: more than 16 prompts");
	for (unsigned p=0;p<E->n_prompt;++p) { tp[p] = toks + (size_t)p*n_tok; nt[p] = (int)n_tok; }
	return clip_encoder_run_ex(E, E->n_prompt, nt, tp, embed, NULL, feat);
}

/* clip_text_encode, src/clip.c:439-488 (one-shot: build, load, run, free), for n_prompt prompts */
MLB_API int clip_text_encode(MLCtx* C, const ClipParams* P, const char* tprefix, unsigned n_prompt, unsigned n_tok,
	const int32_t* toks, float* embed, float* feat, int clip_skip, bool norm, uint64_t synth_seed)
{
	if (n_tok + 2 > (unsigned)P->n_token) return mlsd_set_error(-1, "prompt too long (max: %d)", P->n_token - 2);
	ClipEncoder E;
	int R = clip_encoder_init(&E, C, P, tprefix, n_prompt, clip_skip, norm, feat != NULL);
	if (R > 0) R = mlctx_params_synth(C, synth_seed);
	if (R > 0) R = clip_encoder_run(&E, n_tok, toks, embed, feat);
	clip_encoder_free(&E);
	mlctx_end(C);
	return R;
}

/* ------------------------------------------------------------------ SDXL label assembly (src/mlimgsynth.c:1485-1499,1542-1557) */
static size_t sd_timestep_embedding(unsigned nsteps, const float* steps, unsigned dim, float max_period, float* out)
{
	unsigned half = dim/2;
	for (unsigned i=0; i<half; ++i) {
		float freq = exp(-log(max_period)*i/half);
		for (unsigned s=0; s<nsteps; ++s) {
			out[s*dim+i     ] = cos(steps[s] * freq);
			out[s*dim+i+half] = sin(steps[s] * freq);
		}
	}
	return nsteps * dim;
}

/* label = [pooled feat (n_feat) | emb(h,w) | emb(0,0) | emb(h,w)], 256 dims per scalar */
MLB_API int sdxl_label_build(const float* feat, int n_feat, int width, int height, float* label, int n_label)
{
	if (n_feat + 6*256 != n_label) return mlsd_set_error(-1, "sdxl_label_build: %d + 1536 != %d", n_feat, n_label);
	memcpy(label, feat, (size_t)n_feat*4);
	float *ld = label + n_feat;
	const float hw[2] = {(float)height, (float)width}, zz[2] = {0, 0};
	ld += sd_timestep_embedding(2, hw, 256, 10000, ld);   /* original size */
	ld += sd_timestep_embedding(2, zz, 256, 10000, ld);   /* crop top,left */
	ld += sd_timestep_embedding(2, hw, 256, 10000, ld);   /* target size */
	return 1;
}
