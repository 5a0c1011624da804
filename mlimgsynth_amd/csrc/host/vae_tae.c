/* KL-VAE decoder and TAESD decoder on the MI355X plan builder — re-creation of the reference's
 * src/vae.c:22-74,130-180,318-411 (decoder path) and src/tae.c:17-39,65-92,117-136.
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>

#define T true
#define F false
#define MLN(NAME,X)  mlctx_tensor_add(C, (NAME), (X))

static int64_t rows_of(const MLTensor* t) { return (int64_t)t->n * t->h * t->w; }

/* ------------------------------------------------------------------ parameters (src/vae.c:22-44) */
MLB_API int vae_params_get(const char* model, VaeParams* V)
{
	memset(V, 0, sizeof(*V));
	int m[5] = {1,2,4,4,0};
	V->ch_x=3; V->ch_z=4; V->ch=128; V->n_res=4; V->n_res_blk=2; memcpy(V->ch_mult, m, sizeof(m));
	V->d_embed=4; V->f_down=8;
	if (!strcmp(model,"sd1") || !strcmp(model,"sd2")) V->scale_factor = 0.18215f;
	else if (!strcmp(model,"sdxl")) V->scale_factor = 0.13025f;
	else if (!strcmp(model,"tiny") || !strcmp(model,"tinyxl") || !strcmp(model,"tinyv")) { V->scale_factor = 0.18215f; V->ch = 64; V->n_res_blk = 1; }
	else return mlsd_set_error(-1, "unknown VAE model '%s'", model);
	return 1;
}

/* raw GEMM record on explicit device pointers (attention products of the single-head 2-D attention) */
static void gemm_raw(MLCtx* C, const char* label, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
	const float* bias, const float* bias_m, float* C32, void* C16, int64_t ldc, double flops)
{
	MLOp *op = mlctx_op_new(C, OP_GEMM, label);
	mlsd_gemm_args *g = &op->u.gemm;
	g->A = A; g->lda = lda; g->W_ = B; g->ldb = ldb; g->M = M; g->N = N; g->K = K;
	g->bias = bias; g->bias_m = bias_m; g->C32 = C32; g->ldc32 = ldc; g->C16 = C16; g->ldc16 = ldc;
	op->flops = flops;
}

/* mlb_attn_2d_self, src/vae.c:46-74: single head over d = C, T = h*w tokens per image.
 * d = 512 does not fit the register-resident flash kernel, and this layer is < 6 % of the decoder's
 * FLOPs, so it runs as GEMM (S = Q.K^T, fp32) -> row softmax (fp16 P) -> GEMM (P.V) per image.
 * V^T is produced directly by swapping the operands of the v projection (V^T = Wv . x^T). */
static MLTensor* mlb_attn_2d_self(MLCtx* C, MLTensor* x)
{
	MLTensor *x0 = x;
	mlctx_block_begin(C);
	const int c = x->c, Tn = x->h * x->w, nb = x->n;
	MLTensor *xn = MLN("norm", mlb_groupnorm_ex(C, x, 32, 1e-6f, 0, 0, NULL));
	if (!xn) return NULL;
	/* q, k: 1x1 convs == linears over channels; parameters keep conv shapes [1,1,c,c] */
	MLTensor *q = MLN("q", mlb_conv2d_ex(C, xn, c, 1, 1, 0, 0, T, NULL));
	MLTensor *k = MLN("k", mlb_conv2d_ex(C, xn, c, 1, 1, 0, 0, T, NULL));
	if (!q || !k) return NULL;
	const char *qd = (const char*)mlt_need16(C, q), *kd = (const char*)mlt_need16(C, k);
	/* v: parameters registered like a conv, used as the A operand */
	mlctx_block_begin(C);
	MLParam *vw = mlctx_param_new(C, "weight", MLT_F16, 1, 1, c, c, 1, 0, 0);
	const void *vwd = vw->dev;
	MLParam *vb = mlctx_param_new(C, "bias", MLT_F32, c, 1, 1, 1, 0, 0, 0);
	const float *vbd = (const float*)vb->dev;
	mlctx_named_op(C, "v");
	const size_t sz_vt = (size_t)c * Tn * 2, sz_s = (size_t)Tn * Tn * 4, sz_p = (size_t)Tn * Tn * 2;
	void *vt = mlctx_dalloc(C, sz_vt, 0);
	float *S = (float*)mlctx_dalloc(C, sz_s, 0);
	void *Pm = mlctx_dalloc(C, sz_p, 0);
	MLTensor *a = mlt_new(C, nb, x->h, x->w, c);
	a->sz16 = (size_t)rows_of(x) * c * 2; a->d16 = mlctx_dalloc(C, a->sz16, 0); a->ld16 = c;
	const char *xnd = (const char*)xn->d16;
	for (int b=0; b<nb; ++b) {
		const size_t ro = (size_t)b * Tn * c * 2;   /* byte offset of image b in [rows][c] fp16 tensors */
		gemm_raw(C, "vae_attn_vT", vwd, c, xnd + ro, c, c, Tn, c, NULL, vbd, NULL, vt, Tn, 2.0*c*(double)Tn*c);
		gemm_raw(C, "vae_attn_qk", qd + ro, c, kd + ro, c, Tn, Tn, c, NULL, NULL, S, NULL, Tn, 2.0*Tn*(double)Tn*c);
		MLOp *op = mlctx_op_new(C, OP_SOFTMAX, "softmax_rows");
		op->u.smax.in = S; op->u.smax.ld_in = Tn; op->u.smax.out = Pm; op->u.smax.ld_out = Tn; op->u.smax.rows = Tn; op->u.smax.cols = Tn;
		op->u.smax.scale = (float)(1.0 / sqrt((double)c));
		gemm_raw(C, "vae_attn_pv", Pm, Tn, vt, Tn, Tn, c, Tn, NULL, NULL, NULL, (char*)a->d16 + ro, c, 2.0*Tn*(double)c*Tn);
	}
	mlctx_drelease(C, vt, sz_vt); mlctx_drelease(C, S, sz_s); mlctx_drelease(C, Pm, sz_p);
	mlb_release(C, q); mlb_release(C, k); mlb_release(C, xn);
	MLEpilogue ep = {0}; ep.resid = x0;
	MLTensor *y = MLN("proj_out", mlb_conv2d_ex(C, a, c, 1, 1, 0, 0, T, &ep));
	mlb_release(C, a);
	return y;
}

/* mlb_kl_encoder, src/vae.c:76-118 */
static MLTensor* mlb_kl_encoder(MLCtx* C, MLTensor* x, int ch_out, int ch, int n_res, int n_res_blk, const int* ch_mult)
{
	char name[64];
	mlctx_block_begin(C);
	MLTensor *y;
	x = MLN("conv_in", mlb_conv2d_ex(C, x, ch, 3, 1, 1, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	int ch_blk = ch;
	for (int i=0; i<n_res; ++i) {
		int ch_blk_out = ch * ch_mult[i];
		for (int j=0; j<n_res_blk; ++j) {
			sprintf(name, "down.%d.block.%d", i, j);
			y = MLN(name, mlb_resnet_ex(C, x, NULL, ch_blk_out)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
			ch_blk = ch_blk_out;
		}
		if (i+1 != n_res) {
			sprintf(name, "down.%d.downsample", i);
			y = MLN(name, mlb_downsample(C, x, ch_blk, T)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
		}
	}
	y = MLN("mid.block_1", mlb_resnet_ex(C, x, NULL, ch_blk)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
	y = MLN("mid.attn_1", mlb_attn_2d_self(C, x));             if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
	y = MLN("mid.block_2", mlb_resnet_ex(C, x, NULL, ch_blk)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
	MLTensor *h = MLN("norm_out", mlb_groupnorm_ex(C, x, 32, 1e-6f, 1, 0, NULL));
	mlb_release(C, x);
	x = MLN("conv_out", mlb_conv2d_ex(C, h, ch_out, 3, 1, 1, 0, T, NULL));
	mlb_release(C, h);
	return x;
}

/* mlb_sdvae_encoder, src/vae.c:120-128: image [W,H,3,N] in [-1,1] -> moments [W/8,H/8,2*ch_z,N] (mean | logvar).
 * The [0,1] -> [-1,1] map of sdvae_encoder_pre (src/vae.h:36-40) is applied by the caller's input conversion. */
MLB_API MLTensor* mlb_sdvae_encoder(MLCtx* C, MLTensor* x, const VaeParams* P)
{
	if (x->c != P->ch_x) { mlctx_fail(C, "sdvae_encoder: image must have %d channels", P->ch_x); return NULL; }
	MLTensor *y = MLN("encoder", mlb_kl_encoder(C, x, P->ch_z*2, P->ch, P->n_res, P->n_res_blk, P->ch_mult));
	if (!y) return NULL;
	y = MLN("quant_conv", mlb_conv2d_ex(C, y, P->ch_z*2, 1, 1, 0, 0, T, NULL));
	if (!y || !mlt_need32(C, y)) return NULL;
	return y;
}

MLB_API int sdvae_encode_init(MLCtx* C, const VaeParams* P, unsigned w, unsigned h, unsigned n_batch, MLTensor** t_img)
{
	if (w % P->f_down || h % P->f_down) return mlsd_set_error(-1, "invalid input image shape: %ux%u", w, h);   /* vae.c:241-243 */
	mlctx_begin(C, "VAE encode");
	mlctx_set_tprefix(C, "vae");
	*t_img = mlctx_input_new_img(C, "img", w, h, P->ch_x, n_batch);
	return *t_img ? 1 : -1;
}

MLB_API int sdvae_encode_build(MLCtx* C, const VaeParams* P, MLTensor* t_img)
{
	if (!mlb_sdvae_encoder(C, t_img, P)) return -1;
	return mlctx_prep(C);
}

/* host-boundary encode: img NCHW [n][3][h][w] in [0,1] -> moments NCHW [n][2*ch_z][h/8][w/8] (no sampling) */
MLB_API int sdvae_encode_run(MLCtx* C, MLTensor* t_img, const float* img, float* moments)
{
	const size_t n_in = t_img->in_bytes / 4;
	float *tmp = (float*)malloc(n_in * 4);
	for (size_t i=0;i<n_in;++i) tmp[i] = img[i]*2 - 1;                      /* sdvae_encoder_pre */
	int R = mlctx_input_set(C, t_img, tmp, t_img->in_bytes);
	free(tmp);
	if (R < 0 || mlctx_compute_checked(C) < 0) return -1;
	MLTensor *r = mlctx_result(C);
	const size_t n = (size_t)r->n * r->h * r->w * r->c;
	if (mlctx_output_get(C, r, moments, n*4) < 0) return -1;
	for (size_t i=0;i<n;++i) if (!isfinite(moments[i])) return mlsd_set_error(-1, "NaN found in encoded latent");   /* mlimgsynth.c:1323 */
	return 1;
}

/* mlb_kl_decoder, src/vae.c:130-169 */
static MLTensor* mlb_kl_decoder(MLCtx* C, MLTensor* x, int ch_out, int ch, int n_res, int n_res_blk, const int* ch_mult)
{
	char name[64];
	mlctx_block_begin(C);
	int ch_blk = ch * ch_mult[n_res-1];
	MLTensor *y;
	x = MLN("conv_in", mlb_conv2d_ex(C, x, ch_blk, 3, 1, 1, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	y = MLN("mid.block_1", mlb_resnet_ex(C, x, NULL, ch_blk)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
	y = MLN("mid.attn_1", mlb_attn_2d_self(C, x));             if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
	y = MLN("mid.block_2", mlb_resnet_ex(C, x, NULL, ch_blk)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
	for (int i=n_res-1; i>=0; --i) {
		int ch_blk_out = ch * ch_mult[i];
		for (int j=0; j<n_res_blk+1; ++j) {
			sprintf(name, "up.%d.block.%d", i, j);
			y = MLN(name, mlb_resnet_ex(C, x, NULL, ch_blk_out)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
			ch_blk = ch_blk_out;
		}
		if (i != 0) {
			sprintf(name, "up.%d.upsample", i);
			y = MLN(name, mlb_upsample(C, x, ch_blk)); if (!y || !mlt_need32(C, y)) return NULL; mlb_release(C, x); x = y;
		}
	}
	MLTensor *h = MLN("norm_out", mlb_groupnorm_ex(C, x, 32, 1e-6f, 1, 0, NULL));
	mlb_release(C, x);
	x = MLN("conv_out", mlb_conv2d_ex(C, h, ch_out, 3, 1, 1, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	mlb_release(C, h);
	return x;
}

/* mlb_sdvae_decoder, src/vae.c:171-180.  The 1/scale_factor of ggml_scale is applied while the latent is
 * converted to channels-last fp16 (same rounding point as the reference's F16 im2col). */
MLB_API MLTensor* mlb_sdvae_decoder(MLCtx* C, MLTensor* x, const VaeParams* P)
{
	if (x->c != P->ch_z) { mlctx_fail(C, "sdvae_decoder: latent must have %d channels", P->ch_z); return NULL; }
	if (x->is_input && !x->d16) { if (!x->in_src) x->in_scale0 = 1 / P->scale_factor; }
	else { mlctx_fail(C, "sdvae_decoder: input must be an unconsumed graph input"); return NULL; }
	MLTensor *y = MLN("post_quant_conv", mlb_conv2d_ex(C, x, P->d_embed, 1, 1, 0, 0, T, NULL));
	if (!y) return NULL;
	return MLN("decoder", mlb_kl_decoder(C, y, P->ch_x, P->ch, P->n_res, P->n_res_blk, P->ch_mult));
}

MLB_API int sdvae_decode_init(MLCtx* C, const VaeParams* P, unsigned lw, unsigned lh, unsigned n_batch, MLTensor** t_latent)
{
	mlctx_begin(C, "VAE decode");
	mlctx_set_tprefix(C, "vae");
	*t_latent = mlctx_input_new_img(C, "latent", lw, lh, P->ch_z, n_batch);
	return *t_latent ? 1 : -1;
}

MLB_API int sdvae_decode_build(MLCtx* C, const VaeParams* P, MLTensor* t_latent)
{
	if (!mlb_sdvae_decoder(C, t_latent, P)) return -1;
	return mlctx_prep(C);
}

static int decode_run(MLCtx* C, MLTensor* t_latent, const float* latent, float* img, float mul, float add)
{
	if (mlctx_input_set(C, t_latent, latent, t_latent->in_bytes) < 0) return -1;
	if (mlctx_compute_checked(C) < 0) return -1;
	MLTensor *r = mlctx_result(C);
	const size_t n = (size_t)r->n * r->h * r->w * r->c;
	if (mlctx_output_get(C, r, img, n*4) < 0) return -1;
	for (size_t i=0;i<n;++i) if (!isfinite(img[i])) return mlsd_set_error(-1, "NaN found in decoder output");   /* mlimgsynth.c:1348 */
	if (mul != 1 || add != 0) for (size_t i=0;i<n;++i) img[i] = (img[i] + add) * mul;
	return 1;
}

MLB_API int sdvae_decode_run(MLCtx* C, MLTensor* t_latent, const float* latent, float* img)
{	/* sdvae_decoder_post: (x+1)/2, src/vae.h:43-47 */
	return decode_run(C, t_latent, latent, img, 0.5f, 1.0f);
}

/* ------------------------------------------------------------------ TAESD decoder (src/tae.c) */
static MLTensor* mlb_sdtae_block(MLCtx* C, MLTensor* x, int ch_out)
{	/* :24-39; ch_in == ch_out always, so the 1x1 skip is never instantiated */
	MLTensor *x0 = x;
	mlctx_block_begin(C);
	if (x->c != ch_out) { mlctx_fail(C, "sdtae_block: channel change not implemented"); return NULL; }
	MLEpilogue relu = {0}; relu.act = MLSD_ACT_RELU;
	MLTensor *a = MLN("conv.0", mlb_conv2d_ex(C, x, ch_out, 3, 1, 1, 0, T, &relu));
	MLTensor *b = MLN("conv.2", mlb_conv2d_ex(C, a, ch_out, 3, 1, 1, 0, T, &relu));
	if (!a || !b) return NULL;
	if (!mlt_need32(C, x0)) return NULL;
	MLEpilogue fin = {0}; fin.act = MLSD_ACT_RELU; fin.resid = x0; fin.act_post = 1;   /* relu(conv + x0) */
	MLTensor *y = MLN("conv.4", mlb_conv2d_ex(C, b, ch_out, 3, 1, 1, 0, T, &fin));
	mlb_release(C, a); mlb_release(C, b);
	return y;
}

#define IDX2NAME(I)  (sprintf(name, "%d", (I)), name)

MLB_API MLTensor* mlb_sdtae_decoder(MLCtx* C, MLTensor* x, const SdTaeParams* P)
{	/* :65-92.  3*tanh(x/3) is applied while converting the latent to channels-last fp16 */
	int iblk = 0;
	char name[32];
	mlctx_block_begin(C);
	if (x->is_input && !x->d16) { if (!x->in_src) x->in_mode = 1; }
	else { mlctx_fail(C, "sdtae_decoder: input must be an unconsumed graph input"); return NULL; }
	MLEpilogue relu = {0}; relu.act = MLSD_ACT_RELU;
	MLTensor *y;
	x = MLN(IDX2NAME(iblk++), mlb_conv2d_ex(C, x, P->ch_inner, 3, 1, 1, 0, T, &relu));  iblk++;   /* conv + relu */
	if (!x) return NULL;
	for (int j=0; j<3; ++j) {
		for (int i=0; i<P->n_blk; ++i) {
			y = MLN(IDX2NAME(iblk++), mlb_sdtae_block(C, x, P->ch_inner)); if (!y) return NULL;
			mlt_need32(C, y);   /* next block's residual */
			mlb_release(C, x); x = y;
		}
		iblk++;   /* ggml_upscale, folded into the next conv */
		y = MLN(IDX2NAME(iblk++), mlb_conv2d_ex(C, x, P->ch_inner, 3, 1, 1, 1, F, NULL)); if (!y) return NULL;
		mlt_need32(C, y);
		mlb_release(C, x); x = y;
	}
	y = MLN(IDX2NAME(iblk++), mlb_sdtae_block(C, x, P->ch_inner)); if (!y) return NULL;
	mlb_release(C, x); x = y;
	y = MLN(IDX2NAME(iblk++), mlb_conv2d_ex(C, x, P->ch_x, 3, 1, 1, 0, T, NULL));
	if (!y || !mlt_need32(C, y)) return NULL;
	mlb_release(C, x);
	return y;
}

MLB_API MLTensor* mlb_sdtae_encoder(MLCtx* C, MLTensor* x, const SdTaeParams* P)
{	/* src/tae.c:43-63 */
	int iblk = 0;
	char name[32];
	mlctx_block_begin(C);
	MLTensor *y;
	x = MLN(IDX2NAME(iblk++), mlb_conv2d_ex(C, x, P->ch_inner, 3, 1, 1, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	y = MLN(IDX2NAME(iblk++), mlb_sdtae_block(C, x, P->ch_inner)); if (!y) return NULL;
	mlb_release(C, x); x = y;
	for (int j=0; j<3; ++j) {
		y = MLN(IDX2NAME(iblk++), mlb_conv2d_ex(C, x, P->ch_inner, 3, 2, 1, 0, F, NULL)); if (!y || !mlt_need32(C, y)) return NULL;
		mlb_release(C, x); x = y;
		for (int i=0; i<P->n_blk; ++i) {
			y = MLN(IDX2NAME(iblk++), mlb_sdtae_block(C, x, P->ch_inner)); if (!y) return NULL;
			mlb_release(C, x); x = y;
		}
	}
	y = MLN(IDX2NAME(iblk++), mlb_conv2d_ex(C, x, P->ch_z, 3, 1, 1, 0, T, NULL));
	if (!y || !mlt_need32(C, y)) return NULL;
	mlb_release(C, x);
	return y;
}

static const SdTaeParams g_sdtae_sd1 = { 3, 64, 4, 3 };   /* src/tae.c:17-22 */

MLB_API int sdtae_decode_init(MLCtx* C, unsigned lw, unsigned lh, unsigned n_batch, MLTensor** t_latent)
{
	mlctx_begin(C, "TAE decode");
	mlctx_set_tprefix(C, "tae");
	*t_latent = mlctx_input_new_img(C, "latent", lw, lh, 4, n_batch);
	return *t_latent ? 1 : -1;
}

MLB_API int sdtae_decode_build(MLCtx* C, MLTensor* t_latent)
{
	MLTensor *out = mlb_sdtae_decoder(C, t_latent, &g_sdtae_sd1);
	if (!out) return -1;
	mlctx_tensor_add(C, "decoder.layers", out);   /* src/tae.c:128 */
	return mlctx_prep(C);
}

MLB_API int sdtae_encode_init(MLCtx* C, unsigned w, unsigned h, unsigned n_batch, MLTensor** t_img)
{
	if (w % 8 || h % 8) return mlsd_set_error(-1, "invalid input image shape: %ux%u", w, h);   /* tae.c:100-103 */
	mlctx_begin(C, "TAE encode");
	mlctx_set_tprefix(C, "tae");
	*t_img = mlctx_input_new_img(C, "img", w, h, 3, n_batch);
	return *t_img ? 1 : -1;
}

MLB_API int sdtae_encode_build(MLCtx* C, MLTensor* t_img)
{
	MLTensor *out = mlb_sdtae_encoder(C, t_img, &g_sdtae_sd1);
	if (!out) return -1;
	mlctx_tensor_add(C, "encoder.layers", out);   /* src/tae.c:110 */
	return mlctx_prep(C);
}

/* img NCHW [n][3][h][w] (used as given: the reference's sdtae_encode applies no pre-scaling) -> latent [n][4][h/8][w/8] */
MLB_API int sdtae_encode_run(MLCtx* C, MLTensor* t_img, const float* img, float* latent)
{
	if (mlctx_input_set(C, t_img, img, t_img->in_bytes) < 0 || mlctx_compute_checked(C) < 0) return -1;
	MLTensor *r = mlctx_result(C);
	return mlctx_output_get(C, r, latent, (size_t)r->n * r->h * r->w * r->c * 4);
}

MLB_API int sdtae_decode_run(MLCtx* C, MLTensor* t_latent, const float* latent, float* img)
{
	return decode_run(C, t_latent, latent, img, 1.0f, 0.0f);
}
