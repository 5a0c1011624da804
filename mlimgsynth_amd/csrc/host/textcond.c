/* Text conditioning of the generation driver on resident CLIP towers.
 *
 * Follows mlis_text_cond_encode / mlis_clip_tokens_encode (reference src/mlimgsynth.c:1423-1468,1501-1563):
 *   SD1.x : cond = CLIP-L embedding, clip_skip 1, final norm                                  [77][768]
 *   SDXL  : cond = CLIP-L embedding (clip_skip 2, no norm) || CLIP-bigG embedding (same)      [77][2048]
 *           label = bigG pooled feature (all layers + norm + text_proj) || size embeddings    [2816]
 *           (sd_timestep_embedding of (h,w),(0,0),(h,w), src/mlimgsynth.c:1485-1499,1542-1557)
 *   empty negative prompt on SDXL => uncond zeroed (uncond_empty_zero, :1702-1703); unlabel still computed.
 * The reference rebuilds each CLIP graph and re-uploads its weights per prompt (src/clip.c:458-486); here the
 * towers are built once and stay in HBM.  "tiny"/"tinyxl" use the 3-layer test tower. */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"

struct MLIS_AmdTextCond {
	char model[16];
	int width, height, xl;
	int n_enc;
	MLCtx* ctx[3];
	ClipEncoder enc[3];     /* SD1: [0] ; SDXL: [0] CLIP-L embed, [1] bigG: embed (tap at clip_skip) + pooled feature in one run */
	int n_ctx, n_label;
};

MLB_API void mlis_amd_textcond_destroy(MLIS_AmdTextCond* T)
{
	if (!T) return;
	for (int i=0;i<T->n_enc;++i) { clip_encoder_free(&T->enc[i]); if (T->ctx[i]) mlctx_destroy(T->ctx[i]); }
	free(T);
}

static int g_defer = 0;   /* set around create_ex: towers are built without synthetic weights (the caller loads them) */

static int tower_add(MLIS_AmdTextCond* T, const char* tower, const char* prefix, int clip_skip, bool norm, bool want_feat,
	uint64_t seed, void* stream)
{
	const int i = T->n_enc;
	ClipParams P;
	if (clip_params_get(tower, &P) < 0) return -1;
	T->ctx[i] = mlctx_new(stream);
	if (!T->ctx[i]) return mlsd_set_error(-1, "textcond: mlctx_new failed");
	T->n_enc = i + 1;
	/* built for TWO prompts (prompt + negative prompt go through a tower in one run); want_feat towers also tap the un-normed
	 * hidden state clip_skip layers from the end, so SDXL's bigG runs once for embedding and pooled feature */
	if (clip_encoder_init_ex(&T->enc[i], T->ctx[i], &P, prefix, 2, want_feat ? 1 : clip_skip, norm, want_feat, want_feat ? clip_skip : 0) < 0) return -1;
	if (!g_defer && mlctx_params_synth(T->ctx[i], seed) < 0) return -1;
	return 1;
}

MLB_API MLIS_AmdTextCond* mlis_amd_textcond_create(const char* model, int width, int height, uint64_t weight_seed, void* stream)
{
	return mlis_amd_textcond_create_ex(model, width, height, weight_seed, stream, 0, 0);
}

/* clip_skip 0 = the model's default (1 for SD1, 2 for SD2 / SDXL: src/mlimgsynth.c:757,766,776); the final norm follows
 * UnetParams.clip_norm (SD1/SD2 true, SDXL false: src/unet.c:33,53,75; src/mlimgsynth.c:1511-1512) */
MLB_API MLIS_AmdTextCond* mlis_amd_textcond_create_ex(const char* model, int width, int height, uint64_t weight_seed, void* stream,
	int clip_skip, int defer_weights)
{
	MLIS_AmdTextCond *T = calloc(1, sizeof(*T));
	if (!T) return NULL;
	snprintf(T->model, sizeof(T->model), "%s", model ? model : "");
	T->width = width; T->height = height;
	int rc = -1;
	g_defer = defer_weights;
	if (!strcmp(T->model, "sd1") || !strcmp(T->model, "tiny") || !strcmp(T->model, "tinyv")) {
		rc = tower_add(T, T->model[0] == 's' ? "vit_l" : "tiny", "clip", clip_skip > 0 ? clip_skip : 1, true, false, weight_seed, stream);
	} else if (!strcmp(T->model, "sd2")) {
		rc = tower_add(T, "vit_h", "clip", clip_skip > 0 ? clip_skip : 2, true, false, weight_seed, stream);
	} else if (!strcmp(T->model, "sdxl") || !strcmp(T->model, "tinyxl")) {
		const char *t1 = T->model[0] == 's' ? "vit_l" : "tiny", *t2 = T->model[0] == 's' ? "vit_bigg" : "tiny";
		const int skip = clip_skip > 0 ? clip_skip : 2;
		T->xl = 1;
		rc = tower_add(T, t1, "clip", skip, false, false, weight_seed, stream);
		if (rc > 0) rc = tower_add(T, t2, "clip2", skip, true, true, weight_seed, stream);   /* embedding (tap, clip_skip, no norm) + pooled feature */
	} else mlsd_set_error(-1, "textcond: unknown model '%s'", T->model);
	g_defer = 0;
	if (rc < 0) { mlis_amd_textcond_destroy(T); return NULL; }
	const int d1 = T->enc[0].P.d_embed, d2 = T->xl ? T->enc[1].P.d_embed : 0;
	T->n_ctx = d1 + d2;
	/* tinyxl: adm = 64 + 32 (shrunken size embedding, zeros) so the UNet's label width stays a multiple of 32 */
	T->n_label = !T->xl ? 0 : (!strcmp(T->model, "sdxl") ? d2 + 1536 : d2 + 32);
	return T;
}

MLB_API int mlis_amd_textcond_dims(const MLIS_AmdTextCond* T, int* n_ctx, int* n_label)
{
	if (!T) return -1;
	if (n_ctx) *n_ctx = T->n_ctx;
	if (n_label) *n_label = T->n_label;
	return 1;
}

MLB_API double mlis_amd_textcond_flops(const MLIS_AmdTextCond* T)
{
	double f = 0;
	for (int i=0; T && i<T->n_enc; ++i) { MLCtxInfo I; mlctx_info(T->ctx[i], &I); f += I.flops; }
	return f;
}

MLB_API int mlis_amd_textcond_n_towers(const MLIS_AmdTextCond* T) { return T ? T->n_enc : 0; }
MLB_API MLCtx* mlis_amd_textcond_ctx(MLIS_AmdTextCond* T, int i) { return (T && i >= 0 && i < T->n_enc) ? T->ctx[i] : NULL; }
MLB_API int mlis_amd_textcond_set_size(MLIS_AmdTextCond* T, int width, int height) { T->width = width; T->height = height; return 1; }

/* token weights (prompt emphasis), mlis_clip_tokens_encode src/mlimgsynth.c:1457-1463: rows 1..n_tok of the embedding are
 * multiplied by the weight of their token; the pooled feature / label is computed WITHOUT weights (:1542-1543) */
static void apply_token_weights(float* emb, int d, int n_tok, const float* w)
{
	if (!w) return;
	for (int t=0;t<n_tok;++t) { float *r = emb + (size_t)(t+1)*d; for (int i=0;i<d;++i) r[i] *= w[t]; }
}

MLB_API int mlis_amd_textcond_encode_w(MLIS_AmdTextCond* T, const int32_t* toks, const float* weights, int n_tok, float* cond, float* label)
{
	if (mlis_amd_textcond_encode(T, toks, n_tok, cond, label) < 0) return -1;
	apply_token_weights(cond, T->n_ctx, n_tok, weights);      /* after the concat: the same weight scales both towers' halves */
	return 1;
}

/* n (1 or 2) prompts through every tower in ONE run each; cond[p] [77][n_ctx], label[p] [n_label] */
static int encode_n(MLIS_AmdTextCond* T, int n, const int32_t* const* toks, const int* n_tok, float* const* cond, float* const* label)
{
	static const int32_t none = 0;
	const int32_t *tp[2]; int nt[2];
	for (int p=0;p<n;++p) {
		if (!cond[p] || n_tok[p] < 0 || (n_tok[p] && !toks[p])) return mlsd_set_error(-1, "textcond_encode: bad arguments");
		tp[p] = toks[p] ? toks[p] : &none; nt[p] = n_tok[p];
	}
	const int d1 = T->enc[0].P.d_embed, NT = T->enc[0].P.n_token;
	if (!T->xl) {
		if (n == 1) return clip_encoder_run_ex(&T->enc[0], 1, nt, tp, cond[0], NULL, NULL);
		float *e = (float*)malloc(sizeof(float) * 2 * NT * (size_t)d1);
		int r = e ? clip_encoder_run_ex(&T->enc[0], 2, nt, tp, e, NULL, NULL) : mlsd_set_error(-1, "textcond: out of memory");
		if (r > 0) for (int p=0;p<2;++p) memcpy(cond[p], e + (size_t)p*NT*d1, sizeof(float)*NT*(size_t)d1);
		free(e);
		return r;
	}
	const int d2 = T->enc[1].P.d_embed;
	for (int p=0;p<n;++p) if (!label[p]) return mlsd_set_error(-1, "textcond_encode: SDXL needs a label output");
	float *e1 = (float*)malloc(sizeof(float) * n * NT * (size_t)(d1 + d2) + sizeof(float) * n * (size_t)d2);
	if (!e1) return mlsd_set_error(-1, "textcond: out of memory");
	float *e2 = e1 + (size_t)n*NT*d1, *ft = e2 + (size_t)n*NT*d2;
	int r = clip_encoder_run_ex(&T->enc[0], (unsigned)n, nt, tp, e1, NULL, NULL);
	if (r > 0) r = clip_encoder_run_ex(&T->enc[1], (unsigned)n, nt, tp, NULL, e2, ft);      /* tap = embedding, final = pooled feature */
	for (int p=0; p<n && r>0; ++p) {
		for (int t=0;t<NT;++t) {   /* concat along the embedding axis, src/mlimgsynth.c:1530-1539 */
			memcpy(cond[p] + (size_t)t * (d1 + d2), e1 + ((size_t)p*NT + t) * d1, sizeof(float) * d1);
			memcpy(cond[p] + (size_t)t * (d1 + d2) + d1, e2 + ((size_t)p*NT + t) * d2, sizeof(float) * d2);
		}
		if (!strcmp(T->model, "sdxl")) r = sdxl_label_build(ft + (size_t)p*d2, d2, T->width, T->height, label[p], T->n_label);
		else { memcpy(label[p], ft + (size_t)p*d2, sizeof(float) * d2); memset(label[p] + d2, 0, sizeof(float) * (size_t)(T->n_label - d2)); }
	}
	free(e1);
	return r;
}

MLB_API int mlis_amd_textcond_encode(MLIS_AmdTextCond* T, const int32_t* toks, int n_tok, float* cond, float* label)
{
	if (!T) return mlsd_set_error(-1, "textcond_encode: bad arguments");
	const int32_t *tp[1] = { toks }; float *c[1] = { cond }, *l[1] = { label };
	return encode_n(T, 1, tp, &n_tok, c, l);
}

MLB_API int mlis_amd_textcond_encode_pair(MLIS_AmdTextCond* T, const int32_t* toks, int n_tok, const int32_t* neg, int n_neg,
	float* cond, float* label, float* ncond, float* nlabel)
{
	if (!T) return mlsd_set_error(-1, "textcond_encode: bad arguments");
	const int32_t *tp[2] = { toks, neg }; const int nt[2] = { n_tok, n_neg };
	float *c[2] = { cond, ncond }, *l[2] = { label, nlabel };
	if (encode_n(T, 2, tp, nt, c, l) < 0) return -1;
	if (T->xl && n_neg == 0) memset(ncond, 0, sizeof(float) * 77 * (size_t)T->n_ctx);
	return 1;
}

/* the pair with token weights (prompt emphasis) on both prompts: what mlis_generate needs, one run per tower */
MLB_API int mlis_amd_textcond_encode_pair_w(MLIS_AmdTextCond* T, const int32_t* toks, const float* w, int n_tok,
	const int32_t* neg, const float* nw, int n_neg, float* cond, float* label, float* ncond, float* nlabel)
{
	if (!T) return mlsd_set_error(-1, "textcond_encode: bad arguments");
	const int32_t *tp[2] = { toks, neg }; const int nt[2] = { n_tok, n_neg };
	float *c[2] = { cond, ncond }, *l[2] = { label, nlabel };
	if (encode_n(T, 2, tp, nt, c, l) < 0) return -1;
	apply_token_weights(cond, T->n_ctx, n_tok, w);
	apply_token_weights(ncond, T->n_ctx, n_neg, nw);
	return 1;
}

/* mlis_text_cond_encode for prompt + negative prompt straight into an engine's conditioning inputs (no caller-side staging):
 * what mlis_generate does between :1688 and :1707, as one library call for launchers (bench.py, multi-GPU rank 0) */
MLB_API int mlis_amd_textcond_apply(MLIS_AmdTextCond* T, MLIS_AmdCtx* E, const int32_t* toks, int n_tok, const int32_t* neg, int n_neg)
{
	const size_t nc = (size_t)77 * T->n_ctx, nl = (size_t)(T->n_label > 0 ? T->n_label : 1);
	float *buf = (float*)malloc(sizeof(float) * 2 * (nc + nl));
	if (!buf) return mlsd_set_error(-1, "textcond_apply: out of memory");
	float *cond = buf, *ncond = buf + nc, *label = buf + 2*nc, *nlabel = label + nl;
	int r = mlis_amd_textcond_encode_pair(T, toks, n_tok, neg, n_neg, cond, T->n_label ? label : NULL, ncond, T->n_label ? nlabel : NULL);
	if (r > 0) r = mlis_amd_set_cond(E, cond, T->n_label ? label : NULL, ncond, T->n_label ? nlabel : NULL);
	free(buf);
	return r;
}
