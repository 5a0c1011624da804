/* NN blocks on the MI355X plan builder — the re-creation of the reference's block library
 * (src/mlblock_nn.c:16-253).  Every block records fused HIP launches instead of ggml nodes:
 *
 *   mlb_nn_linear      GEMM (+bias, +activation, +residual)                 src/mlblock_nn.c:16-28
 *   mlb_nn_conv2d      implicit-GEMM conv (+bias, +time-emb, +residual)     :31-55
 *   mlb_nn_groupnorm   stats + apply(+SiLU), concat-aware                   :78-103
 *   mlb_nn_layer_norm  one-wave-per-row kernel                              :58-75
 *   mlb_resnet         GN+SiLU -> conv(+emb) -> GN+SiLU -> conv(+skip)      :129-156
 *   mlb_attn_mhead     fused QKV GEMM -> flash attention -> out_proj(+res)  :190-231
 *   mlb_GEGLU          GEMM with gating epilogue (interleaved weights)      :159-172
 *   mlb_basic_transf   LN/attn/LN/cross-attn/LN/FF with fused residuals     :234-253
 *
 * Parameter naming follows the reference exactly: each block opens a scope with
 * mlctx_block_begin() and registers its parameters by local name ("weight", "bias"); the
 * caller names the block afterwards with mlctx_tensor_add() (the MLN macro of the reference).
 */
#include "mlblock_int.h"
#include <math.h>

#define T true
#define MLN(NAME,X)  mlctx_tensor_add(C, (NAME), (X))

static int64_t rows_of(const MLTensor* t) { return (int64_t)t->n * t->h * t->w; }

/* ------------------------------------------------------------------ linear */
MLTensor* mlb_linear_ex(MLCtx* C, MLTensor* x, int n_out, bool bias, const MLEpilogue* ep, int geglu)
{
	if (!x || C->err) return NULL;
	mlctx_block_begin(C);
	const int n_in = x->c;
	/* C->wtype is the checkpoint's linear weight type (F16 | F32 | BF16).  The device copy is always F16: weights are rounded
	 * once at load time (mlctx_param_set), which is what ggml's F16 path does to the activations anyway; an F32 checkpoint
	 * therefore differs from the reference's pure-fp32 mul_mat by the fp16 rounding of the weights (stated in DESIGN.md). */
	if (C->wtype != MLT_F16 && C->wtype != MLT_F32 && C->wtype != MLT_BF16) { mlctx_fail(C, "unsupported linear weight type %d", C->wtype); return NULL; }
	if (n_in % 8) { mlctx_fail(C, "linear: n_in=%d must be a multiple of 8", n_in); return NULL; }
	const void *xd = mlt_need16(C, x);
	if (!xd) return NULL;
	MLParam *w = mlctx_param_new(C, "weight", MLT_F16, n_in, n_out, 1, 1, geglu ? 2 : 0, 0, 0);
	const void *wd = w->dev;
	const float *bd = NULL;
	if (bias) { MLParam *b = mlctx_param_new(C, "bias", MLT_F32, n_out, 1, 1, 1, geglu ? 3 : 0, 0, 0); bd = (const float*)b->dev; }
	const int n_res = geglu ? n_out/2 : n_out;
	MLTensor *y = mlt_new(C, x->n, x->h, x->w, n_res);
	const float *rd = NULL; int64_t ldr = 0;
	if (ep && ep->resid) {
		if (rows_of(ep->resid) != rows_of(x) || ep->resid->c != n_res) { mlctx_fail(C, "linear: residual shape mismatch"); return NULL; }
		rd = mlt_need32(C, ep->resid); ldr = ep->resid->ld32;
	}
	MLOp *op = mlctx_op_new(C, OP_GEMM, "");
	mlsd_gemm_args *g = &op->u.gemm;
	g->A = xd; g->lda = x->ld16; g->conv = 0;
	g->W_ = wd; g->ldb = n_in; g->M = (int)rows_of(x); g->N = n_out; g->K = n_in;
	g->bias = bd; g->resid = rd; g->ldr = ldr;
	g->act = geglu ? MLSD_ACT_GEGLU : (ep ? ep->act : MLSD_ACT_NONE);
	g->act_after_resid = ep ? ep->act_post : 0;
	op->flops = 2.0 * g->M * (double)n_out * n_in;
	y->prod = C->n_ops - 1;
	return y;
}

MLB_API MLTensor* mlb_nn_linear(MLCtx* C, MLTensor* x, int n_out, bool bias)
{
	return mlb_linear_ex(C, x, n_out, bias, NULL, 0);
}

/* ------------------------------------------------------------------ conv2d */
MLTensor* mlb_conv2d_ex(MLCtx* C, MLTensor* x, int ch_out, int k, int s, int p, int upsample, bool bias, const MLEpilogue* ep)
{
	return mlb_conv2d_ex2(C, x, ch_out, k, s, p, p, upsample, bias, ep);
}

/* p = zero padding before the first row/column, p_end = after the last (ggml_pad(x,1,1,0,0) + conv p=0 of the VAE
 * encoder's downsample, src/mlblock_nn.c:109-111, is p = 0, p_end = 1: the gather's bounds check supplies the zeros) */
MLTensor* mlb_conv2d_ex2(MLCtx* C, MLTensor* x, int ch_out, int k, int s, int p, int p_end, int upsample, bool bias, const MLEpilogue* ep)
{
	if (!x || C->err) return NULL;
	mlctx_block_begin(C);
	const int ch_in = x->c, cpad = (ch_in + 7) / 8 * 8;
	const void *xd = mlt_need16(C, x);
	if (!xd) return NULL;
	if (x->ld16 != cpad && (ch_in % 8)) { mlctx_fail(C, "conv2d: input channels %d need padding to %d", ch_in, cpad); return NULL; }
	/* Warning of the reference kept: conv weights are always F16 (src/mlblock_nn.c:42-43) */
	MLParam *w = mlctx_param_new(C, "weight", MLT_F16, k, k, ch_in, ch_out, 1, 0, 0);
	const void *wd = w->dev;
	const float *bd = NULL;
	if (bias) { MLParam *b = mlctx_param_new(C, "bias", MLT_F32, ch_out, 1, 1, 1, 0, 0, 0); bd = (const float*)b->dev; }
	const int H = x->h, W = x->w, He = upsample ? 2*H : H, We = upsample ? 2*W : W;
	const int OH = (He + p + p_end - k)/s + 1, OW = (We + p + p_end - k)/s + 1;
	MLTensor *y = mlt_new(C, x->n, OH, OW, ch_out);
	const float *rd = NULL, *rb = NULL; int64_t ldr = 0;
	if (ep && ep->resid) {
		if (rows_of(ep->resid) != rows_of(y) || ep->resid->c != ch_out) { mlctx_fail(C, "conv2d: residual shape mismatch"); return NULL; }
		rd = mlt_need32(C, ep->resid); ldr = ep->resid->ld32;
	}
	if (ep && ep->rowbias) {
		if (ep->rowbias->c != ch_out || rows_of(ep->rowbias) != x->n) { mlctx_fail(C, "conv2d: embedding shape mismatch"); return NULL; }
		rb = mlt_need32(C, ep->rowbias);
	}
	MLOp *op = mlctx_op_new(C, OP_GEMM, "");
	mlsd_gemm_args *g = &op->u.gemm;
	g->A = xd; g->lda = x->ld16; g->conv = 1; g->n_img = x->n; g->H = H; g->W = W; g->Cin = cpad; g->OH = OH; g->OW = OW;
	g->KH = k; g->KW = k; g->stride = s; g->pad = p; g->upsample = upsample;
	g->W_ = wd; g->ldb = (int64_t)k*k*cpad; g->M = x->n*OH*OW; g->N = ch_out; g->K = k*k*cpad;
	g->bias = bd; g->rowbias = rb; g->rows_per_batch = OH*OW; g->ldrb = (ep && ep->rowbias) ? ep->rowbias->ld32 : ch_out; g->resid = rd; g->ldr = ldr;
	g->act = ep ? ep->act : MLSD_ACT_NONE;
	g->act_after_resid = ep ? ep->act_post : 0;
	op->flops = 2.0 * g->M * (double)ch_out * k * k * ch_in;
	y->prod = C->n_ops - 1;
	return y;
}

MLB_API MLTensor* mlb_nn_conv2d(MLCtx* C, MLTensor* x, int ch_out,
	int k0, int k1, int s0, int s1, int p0, int p1, int d0, int d1, bool bias)
{
	if (k0 != k1 || s0 != s1 || p0 != p1 || d0 != 1 || d1 != 1) { mlctx_fail(C, "conv2d: only square kernels/strides/pads, dilation 1"); return NULL; }
	return mlb_conv2d_ex(C, x, ch_out, k0, s0, p0, 0, bias, NULL);
}

/* ------------------------------------------------------------------ norms */
MLTensor* mlb_groupnorm_ex(MLCtx* C, MLTensor* x, int n_grp, float eps, int silu, int want_raw16, MLTensor** raw_out)
{
	if (!x || C->err) return NULL;
	mlctx_block_begin(C);
	if (!(eps > 0)) eps = 1e-5f;
	const int Cn = x->c;
	MLParam *w = mlctx_param_new(C, "weight", MLT_F32, Cn, 1, 1, 1, 0, 0, 0);
	const float *wd = (const float*)w->dev;
	MLParam *b = mlctx_param_new(C, "bias", MLT_F32, Cn, 1, 1, 1, 0, 0, 0);
	const float *bd = (const float*)b->dev;
	const MLTensor *a = x->cat_a ? x->cat_a : x, *bb = x->cat_a ? x->cat_b : NULL;
	const float *x1 = mlt_need32(C, (MLTensor*)a), *x2 = bb ? mlt_need32(C, (MLTensor*)bb) : NULL;
	if (!x1 || (bb && !x2)) return NULL;
	const int HW = x->h * x->w;
	MLTensor *y = mlt_new(C, x->n, x->h, x->w, Cn);
	y->sz16 = (size_t)rows_of(x) * Cn * 2;
	y->d16 = mlctx_dalloc(C, y->sz16, 0); y->ld16 = Cn;
	MLTensor *raw = NULL;
	if (want_raw16) {
		raw = mlt_new(C, x->n, x->h, x->w, Cn);
		raw->sz16 = y->sz16; raw->d16 = mlctx_dalloc(C, raw->sz16, 0); raw->ld16 = Cn;
		if (raw_out) *raw_out = raw;
	}
	void *ws = mlctx_dalloc(C, mlsd_groupnorm_ws_bytes(x->n, HW, n_grp), 0);
	MLOp *op = mlctx_op_new(C, OP_GN, silu ? "groupnorm_silu" : "groupnorm");
	mlsd_gn_args *g = &op->u.gn;
	g->x1 = x1; g->ld1 = a->ld32; g->C1 = a->c;
	g->x2 = x2; g->ld2 = bb ? bb->ld32 : 0; g->C2 = bb ? bb->c : 0;
	g->n_img = x->n; g->HW = HW; g->n_grp = n_grp; g->eps = eps; g->gamma = wd; g->beta = bd; g->silu = silu;
	g->y16 = y->d16; g->raw16 = raw ? raw->d16 : NULL; g->ws = ws;
	op->gn_src[0] = a->prod; op->gn_src[1] = bb ? bb->prod : -1;       /* the defining ops of the sources (statistics wiring) */
	return y;
}

MLB_API MLTensor* mlb_nn_groupnorm(MLCtx* C, MLTensor* x, int n_grp, bool affine, float eps)
{
	if (!affine) { mlctx_fail(C, "groupnorm without affine is not implemented"); return NULL; }
	return mlb_groupnorm_ex(C, x, n_grp, eps, 0, 0, NULL);
}

MLTensor* mlb_layer_norm_ex(MLCtx* C, MLTensor* x, float eps, int out32)
{
	if (!x || C->err) return NULL;
	mlctx_block_begin(C);
	if (!(eps > 0)) eps = 1e-5f;
	const int d = x->c;
	MLParam *w = mlctx_param_new(C, "weight", MLT_F32, d, 1, 1, 1, 0, 0, 0);
	const float *wd = (const float*)w->dev;
	MLParam *b = mlctx_param_new(C, "bias", MLT_F32, d, 1, 1, 1, 0, 0, 0);
	const float *bd = (const float*)b->dev;
	const float *xd = mlt_need32(C, x);
	if (!xd) return NULL;
	MLTensor *y = mlt_new(C, x->n, x->h, x->w, d);
	const int64_t rows = rows_of(x);
	y->sz16 = (size_t)rows * d * 2; y->d16 = mlctx_dalloc(C, y->sz16, 0); y->ld16 = d;
	if (out32) { y->sz32 = (size_t)rows * d * 4; y->d32 = (float*)mlctx_dalloc(C, y->sz32, 0); y->ld32 = d; }
	MLOp *op = mlctx_op_new(C, OP_LN, "layernorm");
	op->u.ln.x = xd; op->u.ln.ldx = x->ld32; op->u.ln.rows = (int)rows; op->u.ln.d = d; op->u.ln.eps = eps;
	op->u.ln.g = wd; op->u.ln.b = bd; op->u.ln.y16 = y->d16; op->u.ln.y32 = y->d32;
	op->gn_src[0] = x->prod; op->gn_src[1] = -1;                          /* the defining op of the input (wire_ln_fold) */
	return y;
}

MLB_API MLTensor* mlb_nn_layer_norm(MLCtx* C, MLTensor* x, bool affine, bool bias, float eps)
{
	if (!affine || !bias) { mlctx_fail(C, "layer_norm without affine+bias is not implemented"); return NULL; }
	return mlb_layer_norm_ex(C, x, eps, 1);
}

/* ------------------------------------------------------------------ down/upsample (src/mlblock_nn.c:105-126) */
MLB_API MLTensor* mlb_downsample(MLCtx* C, MLTensor* x, int ch_out, bool vae)
{
	mlctx_block_begin(C);
	if (vae) return MLN("conv", mlb_conv2d_ex2(C, x, ch_out, 3, 2, 0, 1, 0, T, NULL));   /* ggml_pad(x,1,1,0,0) + conv p=0 (:109-111) */
	return MLN("conv", mlb_conv2d_ex(C, x, ch_out, 3, 2, 1, 0, T, NULL));
}

MLB_API MLTensor* mlb_upsample(MLCtx* C, MLTensor* x, int ch_out)
{
	mlctx_block_begin(C);
	/* ggml_upscale(x,2,NEAREST) is folded into the conv's gather */
	return MLN("conv", mlb_conv2d_ex(C, x, ch_out, 3, 1, 1, 1, T, NULL));
}

/* ------------------------------------------------------------------ embedding helpers */
static void* silu16_of(MLCtx* C, MLTensor* emb)
{	/* ggml_silu(emb) of every resnet (src/mlblock_nn.c:140) is computed once per graph */
	if (emb->silu16) return emb->silu16;
	const float *e = mlt_need32(C, emb);
	if (!e) return NULL;
	const size_t n = (size_t)rows_of(emb) * emb->c;
	emb->sz_silu = n * 2;
	emb->silu16 = mlctx_dalloc(C, emb->sz_silu, 0);
	MLOp *op = mlctx_op_new(C, OP_ACT, "silu_f16");
	op->u.act.x = e; op->u.act.y = emb->silu16; op->u.act.n = n; op->u.act.act = MLSD_ACT_SILU;
	return emb->silu16;
}

/* Every resnet of a UNet projects the SAME silu(emb) with its own Linear (emb_proj, src/mlblock_nn.c:140-143): 21 (SDXL) /
 * 22 (SD1.5) launches on 2..8 rows, 13-15 us each, become ONE GEMM [rows x n_total x n_in]; the parameters keep the
 * reference's names and live in consecutive row blocks of one buffer, each conv1 takes its slice as row bias. */
int mlb_emb_proj_batch(MLCtx* C, MLTensor* emb, int n_total)
{
	if (!emb || n_total <= 0 || C->err) return -1;
	void *e16 = silu16_of(C, emb);
	if (!e16) return -1;
	const int n_in = emb->c;
	const int64_t rows = rows_of(emb);
	C->epb.emb = emb; C->epb.n_in = n_in; C->epb.n_total = n_total; C->epb.n_used = 0;
	C->epb.wbase = (char*)mlctx_walloc(C, (size_t)n_total * n_in * 2);
	C->epb.bbase = (float*)mlctx_walloc(C, (size_t)n_total * 4);
	C->epb.out32 = (float*)mlctx_dalloc(C, (size_t)rows * n_total * 4, 0);
	MLOp *op = mlctx_op_new(C, OP_GEMM, "");
	mlsd_gemm_args *g = &op->u.gemm;
	g->A = e16; g->lda = n_in; g->W_ = C->epb.wbase; g->ldb = n_in; g->M = (int)rows; g->N = n_total; g->K = n_in;
	g->bias = C->epb.bbase; g->C32 = C->epb.out32; g->ldc32 = n_total;
	op->flops = 2.0 * rows * (double)n_total * n_in;
	return 1;
}

/* ------------------------------------------------------------------ resnet (src/mlblock_nn.c:129-156) */
MLTensor* mlb_resnet_ex(MLCtx* C, MLTensor* x, MLTensor* emb, int ch_out)
{
	if (!x || C->err) return NULL;
	MLTensor *x0 = x;
	const int ch_in = x->c;
	mlctx_block_begin(C);
	const int need_skip = ch_in != ch_out;
	/* the 1x1 skip conv reads the raw input in fp16; for a virtual concat input (no materialised tensor)
	 * the norm1 kernel emits that copy on the side */
	MLTensor *raw = NULL;
	MLTensor *h = MLN("norm1", mlb_groupnorm_ex(C, x, 32, 1e-6f, 1, need_skip && x->cat_a != NULL, &raw));
	MLEpilogue ep1 = {0};
	MLTensor *ep_t = NULL;
	if (emb) {
		/* emb_proj = Linear(silu(emb)) (:140-143); sibling scopes may be recorded in any order, it is
		 * launched before conv1 because conv1's epilogue adds it per image */
		if (C->epb.emb == emb && emb->c == C->epb.n_in && C->epb.n_used + ch_out <= C->epb.n_total) {
			/* slice of the batched projection (mlb_emb_proj_batch): parameters at their row block, output columns as a view */
			const int off = C->epb.n_used;
			mlctx_block_begin(C);   /* the mlb_nn_linear scope of the reference */
			mlctx_param_new_at(C, "weight", MLT_F16, emb->c, ch_out, 1, 1, 0, C->epb.wbase + (size_t)off * emb->c * 2);
			mlctx_param_new_at(C, "bias", MLT_F32, ch_out, 1, 1, 1, 0, C->epb.bbase + off);
			mlctx_named_op(C, "emb_proj");
			C->epb.n_used += ch_out;
			ep_t = mlt_new(C, emb->n, emb->h, emb->w, ch_out);
			ep_t->d32 = C->epb.out32 + off; ep_t->ld32 = C->epb.n_total; ep_t->sz32 = 0; ep_t->prod = -1;   /* view: not arena owned */
		} else {
			void *e16 = silu16_of(C, emb);
			if (!e16) return NULL;
			MLTensor et = *emb; et.d16 = e16; et.ld16 = emb->c; et.prod = -1;   /* view: silu(emb) in fp16 */
			ep_t = MLN("emb_proj", mlb_linear_ex(C, &et, ch_out, T, NULL, 0));
		}
		ep1.rowbias = ep_t;
	}
	MLTensor *y1 = MLN("conv1", mlb_conv2d_ex(C, h, ch_out, 3, 1, 1, 0, T, &ep1));
	if (!y1 || !mlt_need32(C, y1)) return NULL;
	mlb_release(C, h);
	MLTensor *h2 = MLN("norm2", mlb_groupnorm_ex(C, y1, 32, 1e-6f, 1, 0, NULL));
	mlb_release(C, y1);
	if (ep_t) mlb_release(C, ep_t);
	MLEpilogue ep2 = {0};
	MLTensor *skip = NULL;
	if (need_skip) {
		skip = MLN("skip_conv", mlb_conv2d_ex(C, raw ? raw : x0, ch_out, 1, 1, 0, 0, T, NULL));
		if (!skip || !mlt_need32(C, skip)) return NULL;
		if (raw) mlb_release(C, raw);
		ep2.resid = skip;
	} else ep2.resid = x0;
	MLTensor *y2 = MLN("conv2", mlb_conv2d_ex(C, h2, ch_out, 3, 1, 1, 0, T, &ep2));
	mlb_release(C, h2);
	if (skip) mlb_release(C, skip);   /* recorded after its last reader: stream order keeps it valid */
	return y2;
}

MLB_API MLTensor* mlb_resnet(MLCtx* C, MLTensor* x, MLTensor* emb, int ch_out) { return mlb_resnet_ex(C, x, emb, ch_out); }

/* ------------------------------------------------------------------ GEGLU / feed-forward (src/mlblock_nn.c:159-187) */
MLB_API MLTensor* mlb_GEGLU(MLCtx* C, MLTensor* x, int d_out)
{
	mlctx_block_begin(C);
	if (d_out % 32) { mlctx_fail(C, "GEGLU: d_out=%d must be a multiple of 32", d_out); return NULL; }
	/* proj (+bias) -> chunk -> gelu(gate) * value, all in the GEMM epilogue */
	return MLN("proj", mlb_linear_ex(C, x, d_out*2, T, NULL, 1));
}

static MLTensor* feed_forward_ex(MLCtx* C, MLTensor* x, int d_out, int mult, MLTensor* resid)
{
	mlctx_block_begin(C);
	const int d_inner = x->c * mult;
	MLTensor *h = MLN("net.0", mlb_GEGLU(C, x, d_inner));
	MLEpilogue ep = {0}; ep.resid = resid;
	MLTensor *y = MLN("net.2", mlb_linear_ex(C, h, d_out, T, &ep, 0));
	mlb_release(C, h);
	return y;
}

MLB_API MLTensor* mlb_feed_forward(MLCtx* C, MLTensor* x, int d_out, int mult) { return feed_forward_ex(C, x, d_out, mult, NULL); }

/* ------------------------------------------------------------------ multi-head attention (src/mlblock_nn.c:190-231) */
static void record_attn(MLCtx* C, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
	void* out, int64_t ldo, int nb, int Tq, int Tk, int n_head, int d_head, int causal)
{
	MLOp *op = mlctx_op_new(C, OP_ATTN, "attention");
	mlsd_attn_args *a = &op->u.attn;
	a->q = q; a->k = k; a->v = v; a->out = out; a->ldq = ldq; a->ldk = ldk; a->ldv = ldv; a->ldo = ldo;
	a->bsq = (int64_t)Tq*ldq; a->bsk = (int64_t)Tk*ldk; a->bsv = (int64_t)Tk*ldv; a->bso = (int64_t)Tq*ldo;
	a->n_batch = nb; a->n_head = n_head; a->d_head = d_head; a->Tq = Tq; a->Tk = Tk; a->causal = causal;
	op->flops = 4.0 * nb * (double)Tq * Tk * n_head * d_head;
}

/* three (or two) projections sharing one input as ONE GEMM: parameters keep the reference's names
 * (q_proj/k_proj/v_proj .weight/.bias) but live in consecutive row blocks of one buffer */
static MLTensor* fused_proj(MLCtx* C, MLTensor* x, int d_embed, bool bias, const char* const* names, int n_proj)
{
	const int n_in = x->c;
	const void *xd = mlt_need16(C, x);
	if (!xd) return NULL;
	if (n_in % 8) { mlctx_fail(C, "attention: n_in=%d must be a multiple of 8", n_in); return NULL; }
	char *wbase = (char*)mlctx_walloc(C, (size_t)n_proj * d_embed * n_in * 2);
	float *bbase = bias ? (float*)mlctx_walloc(C, (size_t)n_proj * d_embed * 4) : NULL;
	for (int i=0;i<n_proj;++i) {
		mlctx_block_begin(C);   /* the mlb_nn_linear scope of the reference */
		mlctx_param_new_at(C, "weight", MLT_F16, n_in, d_embed, 1, 1, 0, wbase + (size_t)i * d_embed * n_in * 2);
		if (bias) mlctx_param_new_at(C, "bias", MLT_F32, d_embed, 1, 1, 1, 0, bbase + (size_t)i * d_embed);
		mlctx_named_op(C, names[i]);
	}
	MLTensor *y = mlt_new(C, x->n, x->h, x->w, n_proj * d_embed);
	MLOp *op = mlctx_op_new(C, OP_GEMM, "");
	mlsd_gemm_args *g = &op->u.gemm;
	g->A = xd; g->lda = x->ld16; g->W_ = wbase; g->ldb = n_in; g->M = (int)rows_of(x); g->N = n_proj * d_embed; g->K = n_in;
	g->bias = bbase;
	op->flops = 2.0 * g->M * (double)g->N * n_in;
	y->prod = C->n_ops - 1;
	return y;
}

int mlb_cross_kv_batch(MLCtx* C, MLTensor* ctx, int n_total)
{	/* The context is the same tensor for every cross-attention layer (src/unet.c:130-132), so all
	 * k_proj/v_proj GEMMs share their A operand: 70 launches of 616x2560x2048 (135 TFLOP/s, 100 blocks each)
	 * become one 616 x n_total x 2048 launch that fills the chip. */
	if (!ctx || n_total <= 0 || C->err) return -1;
	const int n_in = ctx->c;
	const int op0 = C->n_ops;
	const void *xd = mlt_need16(C, ctx);
	if (!xd) return -1;
	const int64_t rows = rows_of(ctx);
	C->kvb.ctx = ctx; C->kvb.n_in = n_in; C->kvb.n_total = n_total; C->kvb.n_used = 0;
	C->kvb.wbase = (char*)mlctx_dalloc(C, (size_t)n_total * n_in * 2, 1);
	C->kvb.out16 = (char*)mlctx_dalloc(C, (size_t)rows * n_total * 2, 0);
	MLOp *op = mlctx_op_new(C, OP_GEMM, "");
	mlsd_gemm_args *g = &op->u.gemm;
	g->A = xd; g->lda = ctx->ld16; g->W_ = C->kvb.wbase; g->ldb = n_in; g->M = (int)rows; g->N = n_total; g->K = n_in;
	g->C16 = C->kvb.out16; g->ldc16 = n_total;
	op->flops = 2.0 * rows * (double)n_total * n_in;
	/* The context is also the same for every STEP of a generation (src/mlimgsynth.c:1707-1719 sets it once before the
	 * sampling loop): the ops recorded here (fp16 conversion of the context + this GEMM) depend on nothing else, their outputs
	 * are never recycled (dalloc without release), so mlctx_compute runs them once per conditioning instead of once per
	 * evaluation (the reference's graph recomputes them 40 times per image).  MLSD_NO_HOIST=1 switches this off. */
	if (ctx->is_input) {
		for (int i=op0; i<C->n_ops; ++i) { C->ops[i].once = 1; C->n_once++; }
		ctx->dirty = &C->static_valid;
	}
	return 1;
}

MLTensor* mlb_attn_mhead_ex(MLCtx* C, MLTensor* q, MLTensor* k, MLTensor* v, int d_out, int d_embed, int n_head,
	bool mask, bool bias, bool bias_out, MLTensor* resid)
{
	if (!q || !k || !v || C->err) return NULL;
	const int d_head = d_embed / n_head;
	if (d_head * n_head != d_embed) { mlctx_fail(C, "attention: d_embed %% n_head != 0"); return NULL; }
	mlctx_block_begin(C);
	const int nb = q->n, Tq = q->h * q->w, Tk = k->h * k->w;
	static const char* const n_qkv[] = {"q_proj", "k_proj", "v_proj"};
	static const char* const n_kv[] = {"k_proj", "v_proj"};
	MLTensor *a = mlt_new(C, q->n, q->h, q->w, d_embed);
	a->sz16 = (size_t)rows_of(q) * d_embed * 2; a->d16 = mlctx_dalloc(C, a->sz16, 0); a->ld16 = d_embed;
	if (q == k && k == v) {
		MLTensor *qkv = fused_proj(C, q, d_embed, bias, n_qkv, 3);
		if (!qkv) return NULL;
		const char *p = (const char*)mlt_need16(C, qkv);
		record_attn(C, p, 3*d_embed, p + (size_t)d_embed*2, 3*d_embed, p + (size_t)d_embed*4, 3*d_embed, a->d16, d_embed,
			nb, Tq, Tk, n_head, d_head, mask);
		mlb_release(C, qkv);
	} else {
		if (k != v) { mlctx_fail(C, "attention: k and v must share their input"); return NULL; }
		const int kv_batched = C->kvb.ctx == k && !bias && k->c == C->kvb.n_in && C->kvb.n_used + 2*d_embed <= C->kvb.n_total;
		/* Round 6: the cross attention of the text context (<= 77 keys, d_head 64: every cross attention of SDXL / SD2) runs at the END of its q projection's launch
		 * (mlsd_gemm_args.xa_*, gemm_pp.hpp PP_EPI_XATTN): no attention dispatch, q never reaches HBM.  Decided here from the shapes alone (mlsd_gemm_xattn_fused is a pure
		 * function of them: the tile rule sits above the table); the V operand's transposed, zero-padded copy is one more step-invariant op behind the batched context
		 * projection -- recorded BEFORE the projection so that an eager first evaluation runs it first.  MLSD_XATTN=0 keeps the two launches (A/B, parity test). */
		void *xa_vt = NULL; const char *xa_pk = NULL;
		if (kv_batched && !mask && d_head == 64 && C->kvb.ctx->is_input) {
			mlsd_gemm_args probe; memset(&probe, 0, sizeof(probe));
			probe.A = probe.W_ = (const void*)16; probe.lda = probe.ldb = q->c; probe.M = (int)rows_of(q); probe.N = d_embed; probe.K = q->c;
			probe.xa_k = probe.xa_vt = (const void*)16; probe.xa_out = (void*)16; probe.xa_ldk = C->kvb.n_total; probe.xa_ldo = d_embed; probe.xa_Tq = Tq; probe.xa_Tk = Tk;
			if (!(q->c % 8) && mlsd_gemm_xattn_fused(&probe)) {
				xa_pk = C->kvb.out16 + (size_t)C->kvb.n_used * 2;
				xa_vt = mlctx_dalloc(C, (size_t)nb * d_embed * 96 * 2, 0);      /* never recycled: written once per conditioning */
				MLOp *vo = mlctx_op_new(C, OP_XA_VT, "xattn_pack_vt");
				vo->u.xavt.v = xa_pk + (size_t)d_embed*2; vo->u.xavt.ldv = C->kvb.n_total; vo->u.xavt.n_img = nb; vo->u.xavt.Tk = Tk; vo->u.xavt.N = d_embed; vo->u.xavt.vt = xa_vt;
				vo->once = 1; C->n_once++;
			}
		}
		MLTensor *qp = MLN("q_proj", mlb_linear_ex(C, q, d_embed, bias, NULL, 0));
		if (!qp) return NULL;
		if (kv_batched) {
			/* slices of the batched context projection: parameters registered under the reference's names */
			const int off = C->kvb.n_used, n_in = k->c;
			for (int i=0;i<2;++i) {
				mlctx_block_begin(C);
				mlctx_param_new_at(C, "weight", MLT_F16, n_in, d_embed, 1, 1, 0, C->kvb.wbase + (size_t)(off + i*d_embed) * n_in * 2);
				mlctx_named_op(C, n_kv[i]);
			}
			C->kvb.n_used += 2*d_embed;
			const char *pk = C->kvb.out16 + (size_t)off * 2;
			MLOp *qo = (xa_vt && qp->prod >= 0) ? &C->ops[qp->prod] : NULL;
			if (qo && qo->kind == OP_GEMM) {
				mlsd_gemm_args *g = &qo->u.gemm;
				g->xa_k = pk; g->xa_ldk = C->kvb.n_total; g->xa_vt = xa_vt; g->xa_out = a->d16; g->xa_ldo = d_embed; g->xa_Tq = Tq; g->xa_Tk = Tk;
				if (mlsd_gemm_xattn_fused(g) == 1) qo->flops += 4.0 * nb * (double)Tq * Tk * n_head * d_head;
				else { g->xa_k = NULL; g->xa_vt = NULL; g->xa_out = NULL; qo = NULL; }
			} else qo = NULL;
			if (!qo) {
				const char *pq = (const char*)mlt_need16(C, qp);
				record_attn(C, pq, d_embed, pk, C->kvb.n_total, pk + (size_t)d_embed*2, C->kvb.n_total, a->d16, d_embed,
					nb, Tq, Tk, n_head, d_head, mask);
			}
			mlb_release(C, qp);
		} else {
			MLTensor *kv = fused_proj(C, k, d_embed, bias, n_kv, 2);
			if (!kv) return NULL;
			const char *pq = (const char*)mlt_need16(C, qp), *pk = (const char*)mlt_need16(C, kv);
			record_attn(C, pq, d_embed, pk, 2*d_embed, pk + (size_t)d_embed*2, 2*d_embed, a->d16, d_embed,
				nb, Tq, Tk, n_head, d_head, mask);
			mlb_release(C, qp); mlb_release(C, kv);
		}
	}
	MLEpilogue ep = {0}; ep.resid = resid;
	MLTensor *o = MLN("out_proj", mlb_linear_ex(C, a, d_out, bias_out, &ep, 0));
	mlb_release(C, a);
	return o;
}

MLB_API MLTensor* mlb_attn_mhead(MLCtx* C, MLTensor* q, MLTensor* k, MLTensor* v,
	int d_out, int d_embed, int n_head, bool mask, bool bias, bool bias_out)
{
	return mlb_attn_mhead_ex(C, q, k, v, d_out, d_embed, n_head, mask, bias, bias_out, NULL);
}

/* ------------------------------------------------------------------ transformer block (src/mlblock_nn.c:234-253) */
MLB_API MLTensor* mlb_basic_transf(MLCtx* C, MLTensor* x, MLTensor* c, int d_out, int d_embed, int n_head)
{
	if (!x || C->err) return NULL;
	mlctx_block_begin(C);
	MLTensor *r = x, *n, *y;
	n = MLN("norm1", mlb_layer_norm_ex(C, x, 0, 0));
	y = MLN("attn1", mlb_attn_mhead_ex(C, n, n, n, d_out, d_embed, n_head, false, false, true, r));
	mlb_release(C, n);
	if (!y || !mlt_need32(C, y)) return NULL;
	mlb_release(C, r);
	r = x = y;
	n = MLN("norm2", mlb_layer_norm_ex(C, x, 0, 0));
	y = MLN("attn2", mlb_attn_mhead_ex(C, n, c, c, d_out, d_embed, n_head, false, false, true, r));
	mlb_release(C, n);
	if (!y || !mlt_need32(C, y)) return NULL;
	mlb_release(C, r);
	r = x = y;
	n = MLN("norm3", mlb_layer_norm_ex(C, x, 0, 0));
	y = MLN("ff", feed_forward_ex(C, n, d_out, 4, r));
	mlb_release(C, n);
	if (!y || !mlt_need32(C, y)) return NULL;
	mlb_release(C, r);
	return y;
}

/* ------------------------------------------------------------------ elementwise graph ops */
MLB_API MLTensor* mlb_concat_ch(MLCtx* C, MLTensor* a, MLTensor* b)
{	/* ggml_concat(a,b,2) (src/unet.c:233) as a view: the consumer (groupnorm) reads both sources */
	if (!a || !b || C->err) return NULL;
	if (a->n != b->n || a->h != b->h || a->w != b->w) { mlctx_fail(C, "concat: shape mismatch"); return NULL; }
	if (!mlt_need32(C, a) || !mlt_need32(C, b)) return NULL;
	MLTensor *t = mlt_new(C, a->n, a->h, a->w, a->c + b->c);
	t->cat_a = a; t->cat_b = b;
	return t;
}

MLB_API MLTensor* mlb_timestep_embedding(MLCtx* C, MLTensor* t, int dim, int max_period)
{	/* ggml_timestep_embedding (src/unet.c:150): t holds one timestep per batch element */
	if (!t || C->err) return NULL;
	const int n = (int)(rows_of(t) * t->c);
	MLTensor *y = mlt_new(C, n, 1, 1, dim);
	y->sz16 = (size_t)n * dim * 2; y->d16 = mlctx_dalloc(C, y->sz16, 0); y->ld16 = dim;
	MLOp *op = mlctx_op_new(C, OP_TEMB, "timestep_embedding");
	op->u.temb.t = t->d32; op->u.temb.n = n; op->u.temb.dim = dim; op->u.temb.maxp = (float)max_period; op->u.temb.out = y->d16;
	return y;
}

MLB_API MLTensor* mlb_silu(MLCtx* C, MLTensor* x)
{	/* ggml_silu(_inplace): fp32 in -> fp16 out (every consumer in these graphs is a GEMM) */
	if (!x || C->err) return NULL;
	const float *xd = mlt_need32(C, x);
	if (!xd) return NULL;
	MLTensor *y = mlt_new(C, x->n, x->h, x->w, x->c);
	const size_t n = (size_t)rows_of(x) * x->c;
	y->sz16 = n * 2; y->d16 = mlctx_dalloc(C, y->sz16, 0); y->ld16 = x->c;
	MLOp *op = mlctx_op_new(C, OP_ACT, "silu_f16");
	op->u.act.x = xd; op->u.act.y = y->d16; op->u.act.n = n; op->u.act.act = MLSD_ACT_SILU;
	return y;
}

MLB_API MLTensor* mlb_add(MLCtx* C, MLTensor* a, MLTensor* b)
{	/* ggml_add(a,b) with equal shapes, folded into a's producing GEMM as a residual epilogue.
	 * `a` must be a not-yet-consumed GEMM/conv output (true for every add in the reference's builders). */
	if (!a || !b || C->err) return NULL;
	if (a->prod < 0 || a->d32 || a->d16) { mlctx_fail(C, "mlb_add: first operand must be an unconsumed linear/conv output"); return NULL; }
	MLOp *op = &C->ops[a->prod];
	if (op->u.gemm.resid) { mlctx_fail(C, "mlb_add: producer already has a residual"); return NULL; }
	if (rows_of(a) != rows_of(b) || a->c != b->c) { mlctx_fail(C, "mlb_add: shape mismatch"); return NULL; }
	/* the residual is READ by a's producer (op a->prod): b must be complete before that op runs.  A b recorded later
	 * (e.g. the reference's mlb_resnet order conv2 -> skip_conv -> add, src/mlblock_nn.c:147-154) would be read stale:
	 * refuse it, the caller has to record b first (mlb_resnet_ex does). */
	if (!b->is_input && (b->cat_a || b->def_op >= a->prod)) {
		mlctx_fail(C, "mlb_add: second operand '%s' is produced after the first operand's GEMM (record it earlier)", b->name);
		return NULL;
	}
	const float *bd = mlt_need32(C, b);
	if (!bd) return NULL;
	op = &C->ops[a->prod];
	op->u.gemm.resid = bd; op->u.gemm.ldr = b->ld32;
	return a;
}
