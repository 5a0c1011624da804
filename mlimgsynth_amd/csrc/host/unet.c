/* UNet denoiser on the MI355X plan builder — re-creation of the reference's src/unet.c.
 * Topology, hyper-parameters and parameter names follow the reference line by line; the graph
 * takes a batch N (cond and uncond of every image together), activations are channels-last and
 * the ggml permute/cont/reshape/concat nodes have no counterpart (they are free here).
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>

#define T true
#define MLN(NAME,X)  mlctx_tensor_add(C, (NAME), (X))

/* ------------------------------------------------------------------ hyper-parameters (src/unet.c:21-83) */
MLB_API int unet_params_get(const char* model, UnetParams* U)
{
	memset(U, 0, sizeof(*U));
	U->n_ch_in=4; U->n_ch_out=4; U->n_res_blk=2; U->n_te=1280; U->n_ch=320;
	U->n_step_train=1000; U->sigma_min=0.029167158f; U->sigma_max=14.614641f;
	if (!strcmp(model,"sd1")) {
		int a[4]={4,2,1,0}, m[5]={1,2,4,4,0}, d[5]={1,1,1,1,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_head=8; U->n_ctx=768; U->clip_norm=1;
	} else if (!strcmp(model,"sd2")) {
		int a[4]={4,2,1,0}, m[5]={1,2,4,4,0}, d[5]={1,1,1,1,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->d_head=64; U->n_ctx=1024; U->clip_norm=1; U->vparam=1;
	} else if (!strcmp(model,"sdxl")) {
		int a[4]={4,2,0,0}, m[5]={1,2,4,0,0}, d[5]={1,2,10,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->d_head=64; U->n_ctx=2048; U->ch_adm_in=2816; U->cond_label=1; U->uncond_empty_zero=1;
	} else if (!strcmp(model,"tiny")) {
		int a[4]={2,1,0,0}, m[5]={1,2,0,0,0}, d[5]={1,1,0,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_ch=64; U->n_te=256; U->n_head=2; U->n_ctx=64; U->n_res_blk=1; U->clip_norm=1;
	} else if (!strcmp(model,"tinyv")) {   /* tiny with v-prediction (SD2-style: d_head given, vparam) */
		int a[4]={2,1,0,0}, m[5]={1,2,0,0,0}, d[5]={1,1,0,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_ch=64; U->n_te=256; U->d_head=32; U->n_ctx=64; U->n_res_blk=1; U->clip_norm=1; U->vparam=1;
	} else if (!strcmp(model,"tinyxl")) {
		int a[4]={2,0,0,0}, m[5]={1,2,0,0,0}, d[5]={1,2,0,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_ch=64; U->n_te=256; U->d_head=64; U->n_ctx=128; U->n_res_blk=2; U->ch_adm_in=96;
		U->cond_label=1; U->uncond_empty_zero=1;
	} else return mlsd_set_error(-1, "unknown UNet model '%s'", model);
	return 1;
}

static bool static_vector_in(const int* svec, int v)
{
	for (unsigned i=0; svec[i]; ++i) if (v == svec[i]) return true;
	return false;
}

/* src/unet.c:110-145.  Consumes nothing: x0 stays valid (it is the block's residual). */
static MLTensor* mlb_spatial_transf(MLCtx* C, MLTensor* x, MLTensor* ctx,
	int d_embed, int d_head, int n_head, int n_depth)
{
	MLTensor *x0 = x;
	char name[64];
	mlctx_block_begin(C);
	const int ch_in = x->c;
	if (!n_head)  n_head  = d_embed / d_head;
	if (!d_head)  d_head  = d_embed / n_head;
	if (!d_embed) d_embed = d_head * n_head;

	MLTensor *h = MLN("norm", mlb_groupnorm_ex(C, x, 32, 1e-6f, 0, 0, NULL));
	/* proj_in: 1x1 conv; its output IS the token matrix [N][h*w][d_embed] (no permute needed) */
	x = MLN("proj_in", mlb_conv2d_ex(C, h, d_embed, 1, 1, 0, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	mlb_release(C, h);
	for (int i=0; i<n_depth; ++i) {
		sprintf(name, "transf.%d", i);
		x = MLN(name, mlb_basic_transf(C, x, ctx, d_embed, d_embed, n_head));   /* consumes x */
		if (!x) return NULL;
	}
	MLEpilogue ep = {0}; ep.resid = x0;
	MLTensor *y = MLN("proj_out", mlb_conv2d_ex(C, x, ch_in, 1, 1, 0, 0, T, &ep));
	mlb_release(C, x);
	return y;
}

/* src/unet.c:147-165 */
static MLTensor* mlb_unet__embed(MLCtx* C, MLTensor* time, MLTensor* label, const UnetParams* P)
{
	MLTensor *te = mlb_timestep_embedding(C, time, P->n_ch, 10000);
	MLEpilogue silu = {0}; silu.act = MLSD_ACT_SILU;
	MLTensor *e0 = MLN("time_embed.0", mlb_linear_ex(C, te, P->n_te, T, &silu, 0));
	MLTensor *emb = MLN("time_embed.2", mlb_linear_ex(C, e0, P->n_te, T, NULL, 0));
	if (!emb) return NULL;
	if (P->ch_adm_in && label) {
		MLTensor *l0 = MLN("label_embed.0", mlb_linear_ex(C, label, P->n_te, T, &silu, 0));
		if (!mlt_need32(C, emb)) return NULL;
		MLEpilogue add = {0}; add.resid = emb;       /* emb + label_embed (ggml_add :161) */
		MLTensor *le = MLN("label_embed.2", mlb_linear_ex(C, l0, P->n_te, T, &add, 0));
		mlb_release(C, l0);
		emb = le;
	}
	if (!mlt_need32(C, emb)) return NULL;
	mlb_release(C, te); mlb_release(C, e0);
	return emb;
}

/* total width of all cross-attention k/v projections of the graph (same walk as __in/__mid/__out) */
static int unet_cross_kv_width(const UnetParams* P)
{
	int total = 0, im = 0, ds = 1;
	for (; P->ch_mult[im]; ++im) {
		if (im) ds *= 2;
		if (static_vector_in(P->attn_res, ds)) total += P->n_res_blk * P->transf_depth[im] * 2 * P->n_ch * P->ch_mult[im];          /* in  */
	}
	im--;
	total += P->transf_depth[im] * 2 * P->n_ch * P->ch_mult[im];                                                                  /* mid */
	for (; im >= 0; --im, ds /= 2)
		if (static_vector_in(P->attn_res, ds)) total += (P->n_res_blk + 1) * P->transf_depth[im] * 2 * P->n_ch * P->ch_mult[im];   /* out */
	return total;
}

/* total width of the resnets' time-embedding projections (same walk; every resnet projects emb to its ch_out) */
static int unet_emb_proj_width(const UnetParams* P)
{
	int total = 0, im = 0;
	for (; P->ch_mult[im]; ++im) total += P->n_res_blk * P->n_ch * P->ch_mult[im];       /* in  */
	im--;
	total += 2 * P->n_ch * P->ch_mult[im];                                                /* mid */
	for (; im >= 0; --im) total += (P->n_res_blk + 1) * P->n_ch * P->ch_mult[im];         /* out */
	return total;
}

MLB_API MLTensor* mlb_unet_denoise(MLCtx* C, MLTensor* x, MLTensor* time, MLTensor* ctx, MLTensor* label, const UnetParams* P)
{
	char name[64];
	mlctx_block_begin(C);
	if (mlb_cross_kv_batch(C, ctx, unet_cross_kv_width(P)) < 0) return NULL;
	MLTensor *emb = mlb_unet__embed(C, time, label, P);
	if (!emb) return NULL;
	if (mlb_emb_proj_batch(C, emb, unet_emb_proj_width(P)) < 0) return NULL;

	/* ---- mlb_unet__in, src/unet.c:167-203 */
	MLTensor *stack[40]; int ns = 0;
	x = MLN("in.conv", mlb_conv2d_ex(C, x, P->n_ch, 3, 1, 1, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	stack[ns++] = x;
	int im=0, i_blk=0, ds=1, ch=P->n_ch;
	for (; P->ch_mult[im]; ++im) {
		if (im) {
			ds *= 2; i_blk++;
			sprintf(name, "in.%d.0", i_blk);
			x = MLN(name, mlb_downsample(C, x, ch, false));
			if (!x || !mlt_need32(C, x)) return NULL;
			stack[ns++] = x;
		}
		for (int j=0; j<P->n_res_blk; ++j) {
			i_blk++;
			sprintf(name, "in.%d.0", i_blk);
			ch = P->n_ch * P->ch_mult[im];
			x = MLN(name, mlb_resnet_ex(C, x, emb, ch));
			if (!x || !mlt_need32(C, x)) return NULL;
			if (static_vector_in(P->attn_res, ds)) {
				sprintf(name, "in.%d.1", i_blk);
				MLTensor *y = MLN(name, mlb_spatial_transf(C, x, ctx, ch, P->d_head, P->n_head, P->transf_depth[im]));
				if (!y || !mlt_need32(C, y)) return NULL;
				mlb_release(C, x);
				x = y;
			}
			stack[ns++] = x;
		}
	}

	/* ---- mlb_unet__mid, src/unet.c:205-217 */
	im = 0; while (P->ch_mult[im+1]) im++;
	ch = P->n_ch * P->ch_mult[im];
	{
		MLTensor *y = MLN("mid.0", mlb_resnet_ex(C, x, emb, ch));   /* x stays on the skip stack */
		if (!y || !mlt_need32(C, y)) return NULL;
		x = y;
		y = MLN("mid.1", mlb_spatial_transf(C, x, ctx, ch, P->d_head, P->n_head, P->transf_depth[im]));
		if (!y || !mlt_need32(C, y)) return NULL;
		mlb_release(C, x); x = y;
		y = MLN("mid.2", mlb_resnet_ex(C, x, emb, ch));
		if (!y || !mlt_need32(C, y)) return NULL;
		mlb_release(C, x); x = y;
	}

	/* ---- mlb_unet__out, src/unet.c:219-261 */
	im = 0; ds = 1;
	while (P->ch_mult[im+1]) { im++; ds *= 2; }
	for (int i_oblk=0; im>=0; --im) {
		for (int j=0; j<P->n_res_blk+1; ++j, ++i_oblk) {
			if (ns <= 0) { mlctx_fail(C, "unet: skip stack underflow"); return NULL; }
			MLTensor *hsk = stack[--ns];
			MLTensor *cat = mlb_concat_ch(C, x, hsk);            /* ggml_concat(x, h, 2): zero-copy */
			int i_sub = 0;
			ch = P->n_ch * P->ch_mult[im];
			sprintf(name, "out.%d.%d", i_oblk, i_sub++);
			MLTensor *y = MLN(name, mlb_resnet_ex(C, cat, emb, ch));
			if (!y || !mlt_need32(C, y)) return NULL;
			mlb_release(C, x); mlb_release(C, hsk);
			x = y;
			if (static_vector_in(P->attn_res, ds)) {
				sprintf(name, "out.%d.%d", i_oblk, i_sub++);
				y = MLN(name, mlb_spatial_transf(C, x, ctx, ch, P->d_head, P->n_head, P->transf_depth[im]));
				if (!y || !mlt_need32(C, y)) return NULL;
				mlb_release(C, x); x = y;
			}
			if (im != 0 && j == P->n_res_blk) {
				sprintf(name, "out.%d.%d", i_oblk, i_sub++);
				y = MLN(name, mlb_upsample(C, x, ch));
				if (!y || !mlt_need32(C, y)) return NULL;
				mlb_release(C, x); x = y;
				ds /= 2;
			}
		}
	}
	if (ns != 0) { mlctx_fail(C, "unet: skip stack not empty"); return NULL; }

	MLTensor *h = MLN("out.norm", mlb_groupnorm_ex(C, x, 32, 1e-6f, 1, 0, NULL));
	mlb_release(C, x);
	x = MLN("out.conv", mlb_conv2d_ex(C, h, P->n_ch_out, 3, 1, 1, 0, T, NULL));
	if (!x || !mlt_need32(C, x)) return NULL;
	mlb_release(C, h);
	mlb_release(C, emb);
	return x;
}

/* ------------------------------------------------------------------ sigma tables (src/unet.c:283-334) */
static float g_log_sigmas_sd[1000];

MLB_API void unet_params_init(void)
{
	if (g_log_sigmas_sd[0]) return;
	unsigned n = 1000;
	double linear_start = 0.00085, linear_end = 0.0120,
	       b = sqrt(linear_start), e = sqrt(linear_end),
	       f = (e - b) / (n - 1), alpha_cumprod = 1.0;
	for (unsigned i=0; i<n; ++i) {
		double beta = b + f*i, alpha = 1.0 - beta*beta;
		alpha_cumprod *= alpha;
		double sigma = sqrt((1 - alpha_cumprod) / alpha_cumprod);
		g_log_sigmas_sd[i] = log(sigma);
	}
}

static float linear_interp(unsigned n, const float* vec, float t)
{
	int ti = t;
	if (ti < 0) ti = 0; else if (ti > (int)n-1) ti = (int)n-1;
	float v1 = vec[ti], v2 = ti+1 < (int)n ? vec[ti+1] : v1;
	return v1*(ti+1-t) + v2*(t-ti);
}

/* position where vec crosses v: first index with vec[i] >= v, then the slope of the FOLLOWING interval
 * (BISECT_RIGHT with a comparison that is never 0: src/unet.c:315-322, src/ccommon/bisect.h:38-51) */
static float linear_est(unsigned n, const float* vec, float v)
{
	size_t b = 0, e = n;
	while (b < e) {
		size_t i = (b + e) / 2;
		if (copysign(1, vec[i] - v) < 0) b = i + 1; else e = i;
	}
	size_t idx = b;
	if (idx + 1 >= n) return n - 1;
	float v1 = vec[idx], v2 = vec[idx+1];
	return idx + (v - v1) / (v2 - v1);
}

MLB_API float unet_sigma_to_t(const UnetParams* P, float sigma)
{
	unet_params_init();
	float ls = log(sigma);
	return linear_est(P->n_step_train, g_log_sigmas_sd, ls);
}

MLB_API float unet_t_to_sigma(const UnetParams* P, float t)
{
	unet_params_init();
	float ls = linear_interp(P->n_step_train, g_log_sigmas_sd, t);
	return exp(ls);
}

/* ------------------------------------------------------------------ graph setup / host-boundary run */
MLB_API int unet_denoise_init_n(UnetState* S, MLCtx* C, const UnetParams* P, unsigned lw, unsigned lh, unsigned n_batch)
{
	unet_params_init();
	memset(S, 0, sizeof(*S));
	mlctx_begin(C, "UNet");
	mlctx_set_tprefix(C, "unet");
	S->t_x = mlctx_input_new_img(C, "x", lw, lh, P->n_ch_in, n_batch);
	S->t_t = mlctx_input_new_seq(C, "t", MLT_F32, n_batch, 1, 1);
	S->t_c = mlctx_input_new_seq(C, "c", MLT_F32, P->n_ctx, 77, n_batch);
	if (P->ch_adm_in) S->t_l = mlctx_input_new_seq(C, "l", MLT_F32, P->ch_adm_in, 1, n_batch);
	S->ctx = C; S->par = P; S->lw = lw; S->lh = lh; S->n_batch = n_batch;
	return 1;
}

/* second half of init, split so that callers may bind the x input to a resident latent first */
MLB_API int unet_denoise_build(UnetState* S)
{
	MLCtx *C = S->ctx;
	S->t_out = mlb_unet_denoise(C, S->t_x, S->t_t, S->t_c, S->t_l, S->par);
	if (!S->t_out) return -1;
	if (mlctx_prep(C) < 0) return -1;
	return 1;
}

MLB_API int unet_denoise_run_n(UnetState* S, const float* x, const float* cond, const float* label,
	const float* sigma, float* dx)
{
	MLCtx *C = S->ctx;
	const UnetParams *P = S->par;
	const int N = S->n_batch, hw = S->lw * S->lh, nc = P->n_ch_in;
	const size_t nx = (size_t)N * nc * hw;
	float *xs = (float*)malloc(nx * sizeof(float));
	float *ts = (float*)malloc(N * sizeof(float));
	for (int n=0; n<N; ++n) {
		ts[n] = unet_sigma_to_t(P, sigma[n]);
		float c_in = 1 / sqrt(sigma[n]*sigma[n] + 1);                 /* src/unet.c:470-472 */
		for (size_t i=0; i<(size_t)nc*hw; ++i) xs[(size_t)n*nc*hw + i] = x[(size_t)n*nc*hw + i] * c_in;
	}
	int R = 1;
	if (mlctx_input_set(C, S->t_x, xs, nx*4) < 0) R = -1;
	if (R > 0 && mlctx_input_set(C, S->t_t, ts, N*4) < 0) R = -1;
	if (R > 0 && mlctx_input_set(C, S->t_c, cond, (size_t)N*77*P->n_ctx*4) < 0) R = -1;
	if (R > 0 && S->t_l && mlctx_input_set(C, S->t_l, label, (size_t)N*P->ch_adm_in*4) < 0) R = -1;
	if (R > 0 && mlctx_compute_checked(C) < 0) R = -1;            /* (re-runs on the hand-off-free plan after a timed-out in-launch hand-off) */
	if (R > 0 && mlctx_output_get(C, S->t_out, dx, nx*4) < 0) R = -1;
	S->nfe++;
	if (R > 0) {
		for (size_t i=0; i<nx; ++i) if (!isfinite(dx[i])) { R = mlsd_set_error(-1, "NaN found in UNet output"); break; }   /* :487 */
	}
	if (R > 0 && P->vparam) {                                                                  /* :490-494 */
		for (int n=0; n<N; ++n) {
			float s = sigma[n], c_skip = s / (s*s + 1), c_out = 1 / sqrt(s*s + 1);
			for (size_t i=0; i<(size_t)nc*hw; ++i) { size_t k = (size_t)n*nc*hw + i; dx[k] = dx[k]*c_out + x[k]*c_skip; }
		}
	}
	free(xs); free(ts);
	return R;
}

/* ------------------------------------------------------------------ the reference's signatures (src/unet.h:55-62, src/unet.c:336-388,460-498) */
MLB_API int unet_denoise_init(UnetState* S, MLCtx* C, const UnetParams* P, unsigned lw, unsigned lh, bool split)
{
	(void)split;   /* --unet-split re-uploads half the weights per half-graph to save memory (:390-458): replaced by residency */
	if (unet_denoise_init_n(S, C, P, lw, lh, 1) < 0) return -1;
	return unet_denoise_build(S);
}

MLB_API int unet_denoise_run(UnetState* S, const LocalTensor* x, const LocalTensor* cond, const LocalTensor* label, float sigma, LocalTensor* dx)
{
	const UnetParams *P = S->par;
	if (!x || !cond || !dx || x->n[0] != S->lw || x->n[1] != S->lh || x->n[2] != P->n_ch_in || x->n[3] != 1)
		return mlsd_set_error(-1, "unet_denoise_run: x shape does not match the initialised graph");
	if (cond->n[0] != P->n_ctx || cond->n[1] != 77) return mlsd_set_error(-1, "unet_denoise_run: cond must be [%d,77]", P->n_ctx);
	if (P->ch_adm_in && (!label || label->n[0] != P->ch_adm_in)) return mlsd_set_error(-1, "unet_denoise_run: label must be [%d]", P->ch_adm_in);
	if (dx != x) ltensor_resize(dx, x->n[0], x->n[1], x->n[2], x->n[3]);   /* ltensor_resize_like(dx, x) (:466) */
	return unet_denoise_run_n(S, x->d, cond->d, label ? label->d : NULL, &sigma, dx->d);
}
