/* ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Model graphs restated op by op from the reference builders:
 *   UNet  src/unet.c:110-281     VAE  src/vae.c:46-180     TAE  src/tae.c:24-92
 *   CLIP  src/clip.c:319-437     NN blocks src/mlblock_nn.c:16-253
 * Parameter keys are the dotted names the reference derives in mlctx_load_prep
 * (src/mlblock.c:67-105), e.g. "unet.in.1.1.transf.0.attn1.q_proj.weight".
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

/* ------------------------------------------------------------------ hyper-parameters */
void orc_unet_params_get(const char* model, OrcUnetParams* U)
{
	memset(U, 0, sizeof(*U));
	U->n_ch_in=4; U->n_ch_out=4; U->n_res_blk=2; U->n_te=1280; U->n_ch=320;
	U->n_step_train=1000; U->sigma_min=0.029167158f; U->sigma_max=14.614641f;
	if (!strcmp(model,"sd1")) {            /* g_unet_sd1, src/unet.c:21-39 */
		int a[4]={4,2,1,0}, m[5]={1,2,4,4,0}, d[5]={1,1,1,1,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_head=8; U->n_ctx=768; U->clip_norm=1;
	} else if (!strcmp(model,"sd2")) {     /* g_unet_sd2, src/unet.c:41-60 */
		int a[4]={4,2,1,0}, m[5]={1,2,4,4,0}, d[5]={1,1,1,1,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->d_head=64; U->n_ctx=1024; U->clip_norm=1; U->vparam=1;
	} else if (!strcmp(model,"sdxl")) {    /* g_unet_sdxl, src/unet.c:62-83 */
		int a[4]={4,2,0,0}, m[5]={1,2,4,0,0}, d[5]={1,2,10,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->d_head=64; U->n_ctx=2048; U->ch_adm_in=2816; U->cond_label=1; U->uncond_empty_zero=1;
	} else if (!strcmp(model,"tiny")) {    /* shrunken SD1-like config for fast tests */
		int a[4]={2,1,0,0}, m[5]={1,2,0,0,0}, d[5]={1,1,0,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_ch=64; U->n_te=256; U->n_head=2; U->n_ctx=64; U->n_res_blk=1; U->clip_norm=1;
	} else if (!strcmp(model,"tinyv")) {   /* tiny with v-prediction (SD2-style: d_head given, vparam) */
		int a[4]={2,1,0,0}, m[5]={1,2,0,0,0}, d[5]={1,1,0,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_ch=64; U->n_te=256; U->d_head=32; U->n_ctx=64; U->n_res_blk=1; U->clip_norm=1; U->vparam=1;
	} else if (!strcmp(model,"tinyxl")) {  /* shrunken SDXL-like config (label, d_head, depth 2) */
		int a[4]={2,0,0,0}, m[5]={1,2,0,0,0}, d[5]={1,2,0,0,0};
		memcpy(U->attn_res,a,sizeof(a)); memcpy(U->ch_mult,m,sizeof(m)); memcpy(U->transf_depth,d,sizeof(d));
		U->n_ch=64; U->n_te=256; U->d_head=64; U->n_ctx=128; U->n_res_blk=2; U->ch_adm_in=96;
		U->cond_label=1; U->uncond_empty_zero=1;
	}
}

void orc_vae_params_get(const char* model, OrcVaeParams* V)
{	/* g_vae_sd1 / g_vae_sdxl, src/vae.c:22-44 */
	memset(V, 0, sizeof(*V));
	int m[5]={1,2,4,4,0};
	V->ch_x=3; V->ch_z=4; V->ch=128; V->n_res=4; V->n_res_blk=2; memcpy(V->ch_mult,m,sizeof(m));
	V->d_embed=4; V->f_down=8;
	V->scale_factor = !strcmp(model,"sdxl") ? 0.13025f : 0.18215f;
	if (!strcmp(model,"tiny") || !strcmp(model,"tinyxl") || !strcmp(model,"tinyv")) { V->ch=64; V->n_res_blk=1; }
}

void orc_clip_params_get(const char* model, OrcClipParams* C)
{	/* src/clip.c:23-57 */
	memset(C, 0, sizeof(*C));
	C->n_vocab=49408; C->n_token=77; C->tok_start=49406; C->tok_end=49407;
	if (!strcmp(model,"vit_l"))      { C->d_embed=768;  C->n_interm=3072; C->n_head=12; C->n_layer=12; C->tok_pad=49407; }
	else if (!strcmp(model,"vit_h")) { C->d_embed=1024; C->n_interm=4096; C->n_head=16; C->n_layer=24; C->tok_pad=0; }
	else if (!strcmp(model,"vit_bigg")) { C->d_embed=1280; C->n_interm=5120; C->n_head=20; C->n_layer=32; C->tok_pad=0; }
	else if (!strcmp(model,"tiny"))  { C->n_vocab=1000; C->d_embed=64; C->n_interm=256; C->n_head=2; C->n_layer=3;
	                                   C->tok_start=998; C->tok_end=999; C->tok_pad=999; }
}

/* ------------------------------------------------------------------ naming context */
typedef struct {
	OParams *P;
	char path[512];
	int len[64], depth;
	int wtype;  /* linear weight type, C->c.wtype (default F16, src/mlimgsynth.c:473) */
} Ctx;

static void push(Ctx* C, const char* name)
{
	int l = (int)strlen(C->path);
	C->len[C->depth++] = l;
	snprintf(C->path + l, sizeof(C->path) - l, "%s%s", l ? "." : "", name);
}
static void pop(Ctx* C) { C->path[C->len[--C->depth]] = 0; }

/* optional trace of named layer outputs (debugging aid for tools/make_torch_golden.py --trace) */
static void (*g_trace_cb)(const char* path, const OT* t) = NULL;
ORACLE_API void orc_set_trace(void (*cb)(const char*, const OT*)) { g_trace_cb = cb; }
#define TRACE(C, y) do { if (g_trace_cb) g_trace_cb((C)->path, (y)); } while (0)

static const OParam* par(Ctx* C, const char* name, int type, int64_t n0, int64_t n1, int64_t n2, int64_t n3)
{
	char key[640];
	snprintf(key, sizeof(key), "%s.%s", C->path, name);
	const OParam *p = orc_params_get(C->P, key, type, n0,n1,n2,n3);
	if (!p) { fprintf(stderr, "oracle: missing/mismatched param %s\n", key); abort(); }
	return p;
}

/* Linear weight type of every model built from here on: the reference takes it from the checkpoint unless WEIGHT_TYPE is set (src/mlimgsynth.c:1235-1236, default F16 :473).
 * ORC_F32 = an fp32 checkpoint (BASELINE configs[0]): ggml's mul_mat then multiplies fp32 weights with fp32 activations, nothing is rounded to F16 (orc_linear). */
static int g_linear_wtype = ORC_F16;
ORACLE_API void orc_set_linear_wtype(int type) { g_linear_wtype = type == ORC_F32 ? ORC_F32 : ORC_F16; }
ORACLE_API int orc_get_linear_wtype(void) { return g_linear_wtype; }

/* ------------------------------------------------------------------ NN blocks (src/mlblock_nn.c) */
static OT* nn_linear(Ctx* C, const char* name, const OT* x, int n_out, int bias)
{	/* :16-28 */
	push(C, name);
	const OParam *w = par(C, "weight", C->wtype, x->ne[0], n_out, 1, 1);
	const OParam *b = bias ? par(C, "bias", ORC_F32, n_out, 1, 1, 1) : NULL;
	OT *y = orc_linear(x, w, b);
	TRACE(C, y);
	pop(C);
	return y;
}

static OT* nn_conv2d(Ctx* C, const char* name, const OT* x, int ch_out, int k, int s, int p, int bias)
{	/* :31-55, weight always F16 */
	push(C, name);
	const OParam *w = par(C, "weight", ORC_F16, k, k, x->ne[2], ch_out);
	const OParam *b = bias ? par(C, "bias", ORC_F32, ch_out, 1, 1, 1) : NULL;
	OT *y = orc_conv2d(x, w, b, s, p);
	TRACE(C, y);
	pop(C);
	return y;
}

static OT* nn_layer_norm(Ctx* C, const char* name, const OT* x)
{	/* :58-75 (affine, bias, eps 0 -> 1e-5) */
	push(C, name);
	const OParam *w = par(C, "weight", ORC_F32, x->ne[0], 1,1,1);
	const OParam *b = par(C, "bias", ORC_F32, x->ne[0], 1,1,1);
	OT *y = orc_layer_norm(x, 1e-5f, w, b);
	TRACE(C, y);
	pop(C);
	return y;
}

static OT* nn_groupnorm32(Ctx* C, const char* name, const OT* x)
{	/* mlb_nn_groupnorm32, src/mlblock_nn.h:22-25: 32 groups, affine, eps 1e-6 */
	push(C, name);
	const OParam *w = par(C, "weight", ORC_F32, x->ne[2], 1,1,1);
	const OParam *b = par(C, "bias", ORC_F32, x->ne[2], 1,1,1);
	OT *y = orc_group_norm(x, 32, 1e-6f, w, b);
	TRACE(C, y);
	pop(C);
	return y;
}

static void add_inplace(OT* x, const OT* y)
{
	int64_t n = ot_nel(x);
	for (int64_t i=0;i<n;++i) x->d[i] += y->d[i];
}

static OT* downsample(Ctx* C, const char* name, const OT* x, int ch_out, int vae)
{	/* mlb_downsample :105-116 */
	push(C, name);
	OT *y;
	if (vae) { OT *xp = orc_pad_end(x, 1, 1); y = nn_conv2d(C, "conv", xp, ch_out, 3, 2, 0, 1); ot_free(xp); }
	else y = nn_conv2d(C, "conv", x, ch_out, 3, 2, 1, 1);
	pop(C);
	return y;
}

static OT* upsample(Ctx* C, const char* name, const OT* x, int ch_out)
{	/* mlb_upsample :118-126 */
	push(C, name);
	OT *u = orc_upscale2(x);
	OT *y = nn_conv2d(C, "conv", u, ch_out, 3, 1, 1, 1);
	ot_free(u);
	pop(C);
	return y;
}

static OT* resnet(Ctx* C, const char* name, const OT* x0, const OT* emb, int ch_out)
{	/* mlb_resnet :129-156 */
	push(C, name);
	int ch_in = (int)x0->ne[2];
	OT *x = nn_groupnorm32(C, "norm1", x0);
	orc_silu(x);
	OT *t = nn_conv2d(C, "conv1", x, ch_out, 3, 1, 1, 1); ot_free(x); x = t;
	if (emb) {
		OT *e = ot_from(emb->d, emb->ne[0], emb->ne[1], emb->ne[2], emb->ne[3]);
		orc_silu(e);
		OT *ep = nn_linear(C, "emb_proj", e, ch_out, 1); ot_free(e);
		int64_t HW = x->ne[0]*x->ne[1];
		for (int64_t c=0;c<ch_out;++c) { float v = ep->d[c]; float *xp = x->d + c*HW; for (int64_t i=0;i<HW;++i) xp[i] += v; }
		ot_free(ep);
	}
	t = nn_groupnorm32(C, "norm2", x); ot_free(x); x = t;
	orc_silu(x);
	t = nn_conv2d(C, "conv2", x, ch_out, 3, 1, 1, 1); ot_free(x); x = t;
	if (ch_in != ch_out) {
		OT *sk = nn_conv2d(C, "skip_conv", x0, ch_out, 1, 1, 0, 1);
		add_inplace(x, sk); ot_free(sk);
	} else add_inplace(x, x0);
	pop(C);
	return x;
}

static OT* attn_mhead(Ctx* C, const char* name, const OT* q_in, const OT* k_in, const OT* v_in,
	int d_out, int d_embed, int n_head, int mask, int bias, int bias_out)
{	/* mlb_attn_mhead :190-231 */
	push(C, name);
	OT *q = nn_linear(C, "q_proj", q_in, d_embed, bias);
	OT *k = nn_linear(C, "k_proj", k_in, d_embed, bias);
	OT *v = nn_linear(C, "v_proj", v_in, d_embed, bias);
	OT *a = orc_attention(q, k, v, n_head, mask);
	ot_free(q); ot_free(k); ot_free(v);
	OT *o = nn_linear(C, "out_proj", a, d_out, bias_out);
	ot_free(a);
	pop(C);
	return o;
}

static OT* geglu(Ctx* C, const char* name, const OT* x, int d_out)
{	/* mlb_GEGLU :159-172: proj to 2*d_out, first half = value, second half = gate (ggml_chunk) */
	push(C, name);
	OT *p = nn_linear(C, "proj", x, d_out*2, 1);
	int64_t T = p->ne[1];
	OT *y = ot_new(d_out, T, 1, 1);
	OT *g = ot_new(d_out, T, 1, 1);
	for (int64_t t=0;t<T;++t) memcpy(g->d + t*d_out, p->d + t*2*d_out + d_out, (size_t)d_out*sizeof(float));
	orc_gelu(g);
	for (int64_t t=0;t<T;++t) for (int64_t j=0;j<d_out;++j) y->d[t*d_out+j] = p->d[t*2*d_out+j] * g->d[t*d_out+j];
	ot_free(p); ot_free(g);
	pop(C);
	return y;
}

static OT* feed_forward(Ctx* C, const char* name, const OT* x, int d_out, int mult)
{	/* mlb_feed_forward :175-187 */
	push(C, name);
	OT *h = geglu(C, "net.0", x, (int)x->ne[0]*mult);
	OT *y = nn_linear(C, "net.2", h, d_out, 1);
	ot_free(h);
	pop(C);
	return y;
}

static OT* basic_transf(Ctx* C, const char* name, OT* x, const OT* ctx, int d_embed, int n_head)
{	/* mlb_basic_transf :234-253; consumes x */
	push(C, name);
	OT *n = nn_layer_norm(C, "norm1", x);
	OT *a = attn_mhead(C, "attn1", n, n, n, d_embed, d_embed, n_head, 0, 0, 1); ot_free(n);
	add_inplace(a, x); ot_free(x); x = a;
	n = nn_layer_norm(C, "norm2", x);
	a = attn_mhead(C, "attn2", n, ctx, ctx, d_embed, d_embed, n_head, 0, 0, 1); ot_free(n);
	add_inplace(a, x); ot_free(x); x = a;
	n = nn_layer_norm(C, "norm3", x);
	a = feed_forward(C, "ff", n, d_embed, 4); ot_free(n);
	add_inplace(a, x); ot_free(x); x = a;
	pop(C);
	return x;
}

/* ------------------------------------------------------------------ UNet (src/unet.c) */
static int in_list(const int* l, int v) { for (int i=0; l[i]; ++i) if (l[i]==v) return 1; return 0; }

static OT* spatial_transf(Ctx* C, const char* name, const OT* x0, const OT* ctx,
	int d_embed, int d_head, int n_head, int n_depth)
{	/* mlb_spatial_transf :110-145 */
	push(C, name);
	int w=(int)x0->ne[0], h=(int)x0->ne[1], ch_in=(int)x0->ne[2];
	if (!n_head) n_head = d_embed / d_head;
	OT *x = nn_groupnorm32(C, "norm", x0);
	OT *t = nn_conv2d(C, "proj_in", x, d_embed, 1, 1, 0, 1); ot_free(x);
	x = orc_nchw_to_tokens(t); ot_free(t);
	for (int i=0;i<n_depth;++i) {
		char nm[32]; snprintf(nm, sizeof(nm), "transf.%d", i);
		x = basic_transf(C, nm, x, ctx, d_embed, n_head);
	}
	t = orc_tokens_to_nchw(x, w, h); ot_free(x);
	x = nn_conv2d(C, "proj_out", t, ch_in, 1, 1, 0, 1); ot_free(t);
	add_inplace(x, x0);
	pop(C);
	return x;
}

OT* orc_unet_graph(OParams* P, const char* prefix, const OrcUnetParams* U,
	const OT* x_in, float time, const OT* ctx, const OT* label)
{
	Ctx Cs; memset(&Cs, 0, sizeof(Cs)); Cs.P = P; Cs.wtype = g_linear_wtype;
	Ctx *C = &Cs;
	push(C, prefix);
	char name[64];

	/* mlb_unet__embed :147-165 */
	OT *te = ot_new(U->n_ch, 1, 1, 1);
	orc_timestep_embedding(&time, 1, U->n_ch, 10000, te->d);
	OT *emb = nn_linear(C, "time_embed.0", te, U->n_te, 1); ot_free(te);
	orc_silu(emb);
	OT *t = nn_linear(C, "time_embed.2", emb, U->n_te, 1); ot_free(emb); emb = t;
	if (U->ch_adm_in && label) {
		OT *le = nn_linear(C, "label_embed.0", label, U->n_te, 1);
		orc_silu(le);
		t = nn_linear(C, "label_embed.2", le, U->n_te, 1); ot_free(le);
		add_inplace(emb, t); ot_free(t);
	}

	/* mlb_unet__in :167-203 */
	OT *stack[32]; int ns = 0;
	OT *x = nn_conv2d(C, "in.conv", x_in, U->n_ch, 3, 1, 1, 1);
	stack[ns++] = x;
	int im=0, i_blk=0, ds=1, ch=U->n_ch;
	for (; U->ch_mult[im]; ++im) {
		if (im) {
			ds *= 2; i_blk++;
			snprintf(name, sizeof(name), "in.%d.0", i_blk);
			x = downsample(C, name, x, ch, 0);
			stack[ns++] = x;
		}
		for (int j=0; j<U->n_res_blk; ++j) {
			i_blk++;
			snprintf(name, sizeof(name), "in.%d.0", i_blk);
			ch = U->n_ch * U->ch_mult[im];
			x = resnet(C, name, x, emb, ch);   /* previous x stays alive on the skip stack */
			if (in_list(U->attn_res, ds)) {
				snprintf(name, sizeof(name), "in.%d.1", i_blk);
				t = spatial_transf(C, name, x, ctx, ch, U->d_head, U->n_head, U->transf_depth[im]);
				ot_free(x); x = t;
			}
			stack[ns++] = x;
		}
	}

	/* mlb_unet__mid :205-217 */
	im = 0; while (U->ch_mult[im+1]) im++;
	ch = U->n_ch * U->ch_mult[im];
	x = resnet(C, "mid.0", x, emb, ch);   /* input x is stack top: not freed here */
	t = spatial_transf(C, "mid.1", x, ctx, ch, U->d_head, U->n_head, U->transf_depth[im]); ot_free(x); x = t;
	t = resnet(C, "mid.2", x, emb, ch); ot_free(x); x = t;

	/* mlb_unet__out :219-261 */
	im = 0; ds = 1; while (U->ch_mult[im+1]) { im++; ds *= 2; }
	for (int i_oblk=0; im>=0; --im) {
		for (int j=0; j<U->n_res_blk+1; ++j, ++i_oblk) {
			OT *h = stack[--ns];
			t = orc_concat_ch(x, h); ot_free(x); ot_free(h); x = t;
			int i_sub = 0;
			ch = U->n_ch * U->ch_mult[im];
			snprintf(name, sizeof(name), "out.%d.%d", i_oblk, i_sub++);
			t = resnet(C, name, x, emb, ch); ot_free(x); x = t;
			if (in_list(U->attn_res, ds)) {
				snprintf(name, sizeof(name), "out.%d.%d", i_oblk, i_sub++);
				t = spatial_transf(C, name, x, ctx, ch, U->d_head, U->n_head, U->transf_depth[im]); ot_free(x); x = t;
			}
			if (im != 0 && j == U->n_res_blk) {
				snprintf(name, sizeof(name), "out.%d.%d", i_oblk, i_sub++);
				t = upsample(C, name, x, ch); ot_free(x); x = t;
				ds /= 2;
			}
		}
	}
	t = nn_groupnorm32(C, "out.norm", x); ot_free(x); x = t;
	orc_silu(x);
	t = nn_conv2d(C, "out.conv", x, U->n_ch_out, 3, 1, 1, 1); ot_free(x); x = t;
	ot_free(emb);
	return x;
}

/* ------------------------------------------------------------------ VAE decoder (src/vae.c) */
static OT* attn_2d_self(Ctx* C, const char* name, const OT* x0)
{	/* mlb_attn_2d_self :46-74: single head over d = C */
	push(C, name);
	int w=(int)x0->ne[0], h=(int)x0->ne[1], c=(int)x0->ne[2];
	OT *x = nn_groupnorm32(C, "norm", x0);
	OT *q4 = nn_conv2d(C, "q", x, c, 1, 1, 0, 1);
	OT *k4 = nn_conv2d(C, "k", x, c, 1, 1, 0, 1);
	OT *v4 = nn_conv2d(C, "v", x, c, 1, 1, 0, 1);
	ot_free(x);
	OT *q = orc_nchw_to_tokens(q4), *k = orc_nchw_to_tokens(k4), *v = orc_nchw_to_tokens(v4);
	ot_free(q4); ot_free(k4); ot_free(v4);
	OT *a = orc_attention(q, k, v, 1, 0);
	ot_free(q); ot_free(k); ot_free(v);
	OT *a4 = orc_tokens_to_nchw(a, w, h); ot_free(a);
	x = nn_conv2d(C, "proj_out", a4, c, 1, 1, 0, 1); ot_free(a4);
	add_inplace(x, x0);
	pop(C);
	return x;
}

OT* orc_vae_decode(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* latent)
{
	Ctx Cs; memset(&Cs, 0, sizeof(Cs)); Cs.P = P; Cs.wtype = g_linear_wtype;
	Ctx *C = &Cs;
	push(C, prefix);
	char name[64];
	/* mlb_sdvae_decoder :171-180 */
	OT *x = ot_from(latent->d, latent->ne[0], latent->ne[1], latent->ne[2], latent->ne[3]);
	{ float f = 1 / V->scale_factor; int64_t n = ot_nel(x); for (int64_t i=0;i<n;++i) x->d[i] *= f; }
	OT *t = nn_conv2d(C, "post_quant_conv", x, V->d_embed, 1, 1, 0, 1); ot_free(x); x = t;
	/* mlb_kl_decoder :130-169 */
	push(C, "decoder");
	int ch_blk = V->ch * V->ch_mult[V->n_res-1];
	t = nn_conv2d(C, "conv_in", x, ch_blk, 3, 1, 1, 1); ot_free(x); x = t;
	t = resnet(C, "mid.block_1", x, NULL, ch_blk); ot_free(x); x = t;
	t = attn_2d_self(C, "mid.attn_1", x); ot_free(x); x = t;
	t = resnet(C, "mid.block_2", x, NULL, ch_blk); ot_free(x); x = t;
	for (int i=V->n_res-1; i>=0; --i) {
		int ch_out = V->ch * V->ch_mult[i];
		for (int j=0; j<V->n_res_blk+1; ++j) {
			snprintf(name, sizeof(name), "up.%d.block.%d", i, j);
			t = resnet(C, name, x, NULL, ch_out); ot_free(x); x = t;
			ch_blk = ch_out;
		}
		if (i != 0) {
			snprintf(name, sizeof(name), "up.%d.upsample", i);
			t = upsample(C, name, x, ch_blk); ot_free(x); x = t;
		}
	}
	t = nn_groupnorm32(C, "norm_out", x); ot_free(x); x = t;
	orc_silu(x);
	t = nn_conv2d(C, "conv_out", x, V->ch_x, 3, 1, 1, 1); ot_free(x); x = t;
	pop(C);
	/* sdvae_decoder_post, src/vae.h:43-47 */
	{ int64_t n = ot_nel(x); for (int64_t i=0;i<n;++i) x->d[i] = (x->d[i]+1)/2; }
	return x;
}

/* sdvae_encode without tiling, src/vae.c:76-128,231-316: image [W,H,3,1] in [0,1] -> moments [W/8,H/8,2*ch_z,1]
 * (mean | logvar) after quant_conv; no sampling (orc_latent_sample) */
OT* orc_vae_encode_moments(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* img)
{
	Ctx Cs; memset(&Cs, 0, sizeof(Cs)); Cs.P = P; Cs.wtype = g_linear_wtype;
	Ctx *C = &Cs;
	push(C, prefix);
	char name[64];
	OT *x = ot_from(img->d, img->ne[0], img->ne[1], img->ne[2], img->ne[3]);
	{ int64_t n = ot_nel(x); for (int64_t i=0;i<n;++i) x->d[i] = x->d[i]*2 - 1; }   /* sdvae_encoder_pre, src/vae.h:36-40 */
	/* mlb_kl_encoder :76-118 */
	push(C, "encoder");
	OT *t = nn_conv2d(C, "conv_in", x, V->ch, 3, 1, 1, 1); ot_free(x); x = t;
	int ch_blk = V->ch;
	for (int i=0; i<V->n_res; ++i) {
		int ch_out = V->ch * V->ch_mult[i];
		for (int j=0; j<V->n_res_blk; ++j) {
			snprintf(name, sizeof(name), "down.%d.block.%d", i, j);
			t = resnet(C, name, x, NULL, ch_out); ot_free(x); x = t;
			ch_blk = ch_out;
		}
		if (i+1 != V->n_res) {
			snprintf(name, sizeof(name), "down.%d.downsample", i);
			t = downsample(C, name, x, ch_blk, 1); ot_free(x); x = t;
		}
	}
	t = resnet(C, "mid.block_1", x, NULL, ch_blk); ot_free(x); x = t;
	t = attn_2d_self(C, "mid.attn_1", x); ot_free(x); x = t;
	t = resnet(C, "mid.block_2", x, NULL, ch_blk); ot_free(x); x = t;
	t = nn_groupnorm32(C, "norm_out", x); ot_free(x); x = t;
	orc_silu(x);
	t = nn_conv2d(C, "conv_out", x, V->ch_z*2, 3, 1, 1, 1); ot_free(x); x = t;
	pop(C);
	/* mlb_sdvae_encoder :120-128 */
	t = nn_conv2d(C, "quant_conv", x, V->ch_z*2, 1, 1, 0, 1); ot_free(x); x = t;
	return x;
}

/* ltensor_copy_slice2 on [W,H,C,1] tensors (src/localtensor.c:15-60) */
static void copy_slice2(OT* dst, const OT* src, int n0, int n1, int di0, int di1, int si0, int si1)
{
	const int64_t C = src->ne[2];
	for (int64_t c=0;c<C;++c) for (int y=0;y<n1;++y) for (int x=0;x<n0;++x)
		dst->d[(c*dst->ne[1] + di1 + y)*dst->ne[0] + di0 + x] = src->d[(c*src->ne[1] + si1 + y)*src->ne[0] + si0 + x];
}

/* sdvae_decode with tile_px > 0, src/vae.c:333-391 */
OT* orc_vae_decode_tiled(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* latent, int tile_px)
{
	const int f = V->f_down, k = 8, lat_n0 = (int)latent->ne[0], lat_n1 = (int)latent->ne[1];
	int n0 = lat_n0, n1 = lat_n1;
	if (tile_px > 0) {
		tile_px = ((tile_px + 63) / 64) * 64;
		n0 = tile_px/f + k*2 < lat_n0 ? tile_px/f + k*2 : lat_n0;
		n1 = tile_px/f + k*2 < lat_n1 ? tile_px/f + k*2 : lat_n1;
		if (n0 == lat_n0 && n1 == lat_n1) tile_px = 0;
	}
	if (tile_px <= 0) return orc_vae_decode(P, prefix, V, latent);
	const int step0 = n0 - k*2, step1 = n1 - k*2;
	const int n_tile0 = (lat_n0 + step0 - 1) / step0, n_tile1 = (lat_n1 + step1 - 1) / step1;
	OT *img = ot_new(lat_n0*f, lat_n1*f, 3, 1), *ltmp = ot_new(n0, n1, 4, 1);
	for (int t1=0; t1<n_tile1; ++t1) {
		int i1 = t1*step1 < lat_n1 - n1 ? t1*step1 : lat_n1 - n1;
		for (int t0=0; t0<n_tile0; ++t0) {
			int i0 = t0*step0 < lat_n0 - n0 ? t0*step0 : lat_n0 - n0;
			copy_slice2(ltmp, latent, n0, n1, 0, 0, i0, i1);
			OT *t = orc_vae_decode(P, prefix, V, ltmp);     /* includes the (x+1)/2 post, elementwise: commutes with the paste */
			int d0 = i0 ? k : 0, d1 = i1 ? k : 0;
			copy_slice2(img, t, (n0-k)*f, (n1-k)*f, (i0+d0)*f, (i1+d1)*f, d0*f, d1*f);
			ot_free(t);
		}
	}
	ot_free(ltmp);
	return img;
}

/* sdvae_encode with tile_px > 0, src/vae.c:231-316 (moments, before sampling) */
OT* orc_vae_encode_moments_tiled(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* img, int tile_px)
{
	const int f = V->f_down, k = f*8, img_n0 = (int)img->ne[0], img_n1 = (int)img->ne[1];
	int n0 = img_n0, n1 = img_n1;
	if (tile_px > 0) {
		tile_px = ((tile_px + 63) / 64) * 64;
		n0 = tile_px + k*2 < img_n0 ? tile_px + k*2 : img_n0;
		n1 = tile_px + k*2 < img_n1 ? tile_px + k*2 : img_n1;
		if (n0 == img_n0 && n1 == img_n1) tile_px = 0;
	}
	if (tile_px <= 0) return orc_vae_encode_moments(P, prefix, V, img);
	const int step0 = n0 - k*2, step1 = n1 - k*2;
	const int n_tile0 = (img_n0 + step0 - 1) / step0, n_tile1 = (img_n1 + step1 - 1) / step1;
	OT *mom = ot_new(img_n0/f, img_n1/f, V->ch_z*2, 1), *itmp = ot_new(n0, n1, 3, 1);
	for (int t1=0; t1<n_tile1; ++t1) {
		int i1 = t1*step1 < img_n1 - n1 ? t1*step1 : img_n1 - n1;
		for (int t0=0; t0<n_tile0; ++t0) {
			int i0 = t0*step0 < img_n0 - n0 ? t0*step0 : img_n0 - n0;
			copy_slice2(itmp, img, n0, n1, 0, 0, i0, i1);
			OT *t = orc_vae_encode_moments(P, prefix, V, itmp);
			int d0 = i0 ? k : 0, d1 = i1 ? k : 0;
			copy_slice2(mom, t, (n0-k)/f, (n1-k)/f, (i0+d0)/f, (i1+d1)/f, d0/f, d1/f);
			ot_free(t);
		}
	}
	ot_free(itmp);
	return mom;
}

/* sdvae_latent_sample / sdvae_latent_mean, src/vae.c:188-229: moments [W,H,2cz,1] -> latent [W,H,cz,1];
 * rnd = n floats of N(0,1) (one rng_randn call) or NULL for the mean */
OT* orc_latent_sample(const OT* mom, const OrcVaeParams* V, const float* rnd)
{
	const int64_t n = mom->ne[0]*mom->ne[1]*(mom->ne[2]/2);
	OT *lat = ot_new(mom->ne[0], mom->ne[1], mom->ne[2]/2, 1);
	const float *mean = mom->d, *logvar = mom->d + n;
	for (int64_t i=0;i<n;++i) {
		float v = mean[i];
		if (rnd) {
			float lv = logvar[i]; lv = lv < -30 ? -30 : (lv > 20 ? 20 : lv);
			v = mean[i] + exp(lv * 0.5) * rnd[i];
		}
		lat->d[i] = v * V->scale_factor;
	}
	return lat;
}

/* ------------------------------------------------------------------ TAE (src/tae.c) */
static OT* tae_block(Ctx* C, int idx, const OT* x0, int ch_out);

OT* orc_tae_encode(OParams* P, const char* prefix, const OT* img)
{	/* mlb_sdtae_encoder :43-63 (the image is used as given: sdtae_encode applies no pre-scaling, :96-115) */
	const int ch_inner=64, ch_z=4, n_blk=3;
	Ctx Cs; memset(&Cs, 0, sizeof(Cs)); Cs.P = P; Cs.wtype = g_linear_wtype;
	Ctx *C = &Cs;
	push(C, prefix); push(C, "encoder.layers");
	char name[16];
	int iblk = 0;
	OT *x = ot_from(img->d, img->ne[0], img->ne[1], img->ne[2], img->ne[3]);
	snprintf(name, sizeof(name), "%d", iblk++);
	OT *t = nn_conv2d(C, name, x, ch_inner, 3, 1, 1, 1); ot_free(x); x = t;
	t = tae_block(C, iblk++, x, ch_inner); ot_free(x); x = t;
	for (int j=0;j<3;++j) {
		snprintf(name, sizeof(name), "%d", iblk++);
		t = nn_conv2d(C, name, x, ch_inner, 3, 2, 1, 0); ot_free(x); x = t;
		for (int i=0;i<n_blk;++i) { t = tae_block(C, iblk++, x, ch_inner); ot_free(x); x = t; }
	}
	snprintf(name, sizeof(name), "%d", iblk++);
	t = nn_conv2d(C, name, x, ch_z, 3, 1, 1, 1); ot_free(x); x = t;
	return x;
}

/* ------------------------------------------------------------------ TAE decoder (src/tae.c) */
static OT* tae_block(Ctx* C, int idx, const OT* x0, int ch_out)
{	/* mlb_sdtae_block :24-39 (ch_in == ch_out always: no skip conv) */
	char name[16]; snprintf(name, sizeof(name), "%d", idx);
	push(C, name);
	OT *x = nn_conv2d(C, "conv.0", x0, ch_out, 3, 1, 1, 1); orc_relu(x);
	OT *t = nn_conv2d(C, "conv.2", x, ch_out, 3, 1, 1, 1); ot_free(x); x = t; orc_relu(x);
	t = nn_conv2d(C, "conv.4", x, ch_out, 3, 1, 1, 1); ot_free(x); x = t;
	add_inplace(x, x0);
	orc_relu(x);
	pop(C);
	return x;
}

OT* orc_tae_decode(OParams* P, const char* prefix, const OT* latent)
{	/* mlb_sdtae_decoder :65-92; params g_sdtae_sd1 :17-22 */
	const int ch_inner=64, ch_x=3, n_blk=3;
	Ctx Cs; memset(&Cs, 0, sizeof(Cs)); Cs.P = P; Cs.wtype = g_linear_wtype;
	Ctx *C = &Cs;
	push(C, prefix); push(C, "decoder.layers");
	char name[16];
	int iblk = 0;
	OT *x = ot_from(latent->d, latent->ne[0], latent->ne[1], latent->ne[2], latent->ne[3]);
	{ int64_t n = ot_nel(x); for (int64_t i=0;i<n;++i) { float v = x->d[i]*(1.0f/3.0f); v = tanhf(v); x->d[i] = v*3.0f; } }
	snprintf(name, sizeof(name), "%d", iblk++);
	OT *t = nn_conv2d(C, name, x, ch_inner, 3, 1, 1, 1); ot_free(x); x = t;
	orc_relu(x); iblk++;
	for (int j=0;j<3;++j) {
		for (int i=0;i<n_blk;++i) { t = tae_block(C, iblk++, x, ch_inner); ot_free(x); x = t; }
		t = orc_upscale2(x); ot_free(x); x = t; iblk++;
		snprintf(name, sizeof(name), "%d", iblk++);
		t = nn_conv2d(C, name, x, ch_inner, 3, 1, 1, 0); ot_free(x); x = t;
	}
	t = tae_block(C, iblk++, x, ch_inner); ot_free(x); x = t;
	snprintf(name, sizeof(name), "%d", iblk++);
	t = nn_conv2d(C, name, x, ch_x, 3, 1, 1, 1); ot_free(x); x = t;
	return x;
}

/* ------------------------------------------------------------------ CLIP text (src/clip.c) */
OT* orc_clip_text_encode(OParams* P, const char* prefix, const OrcClipParams* K,
	const int32_t* tokens, int clip_skip, int norm, int want_feat, int i_tok_end)
{
	Ctx Cs; memset(&Cs, 0, sizeof(Cs)); Cs.P = P; Cs.wtype = g_linear_wtype;
	Ctx *C = &Cs;
	if (want_feat) { clip_skip = -1; norm = 1; }   /* clip.c:446 */
	push(C, prefix); push(C, "text");
	const int d = K->d_embed, T = K->n_token;
	/* mlb_clip_embeddings :319-344 */
	push(C, "embed");
	const OParam *tw = par(C, "token.weight", C->wtype, d, K->n_vocab, 1, 1);
	const OParam *pw = par(C, "position.weight", ORC_F32, d, T, 1, 1);
	pop(C);
	OT *x = ot_new(d, T, 1, 1);
	for (int t=0;t<T;++t) for (int i=0;i<d;++i) x->d[t*d+i] = tw->d[(int64_t)tokens[t]*d + i] + pw->d[t*d+i];
	/* mlb_clip_encoder / mlb_clip_layer :362-393 */
	int n_layer = K->n_layer;
	if (clip_skip > 1) n_layer -= clip_skip-1;
	push(C, "encoder");
	for (int l=0;l<n_layer;++l) {
		char name[32]; snprintf(name, sizeof(name), "layers.%d", l);
		push(C, name);
		OT *n = nn_layer_norm(C, "norm1", x);
		OT *a = attn_mhead(C, "attn", n, n, n, d, d, K->n_head, 1, 1, 1); ot_free(n);
		add_inplace(a, x); ot_free(x); x = a;
		n = nn_layer_norm(C, "norm2", x);
		push(C, "mlp");   /* mlb_clip_mlp :346-360 */
		OT *h = nn_linear(C, "fc1", n, K->n_interm, 1); ot_free(n);
		if (d == 1024 || d == 1280) orc_gelu(h); else orc_gelu_quick(h);
		OT *o = nn_linear(C, "fc2", h, d, 1); ot_free(h);
		pop(C);
		add_inplace(o, x); ot_free(x); x = o;
		pop(C);
	}
	pop(C);
	if (norm) { OT *t = nn_layer_norm(C, "ln_final", x); ot_free(x); x = t; }
	if (want_feat) {
		/* mlb_clip_text_proj :418-437: feat = text_proj^T . x[:, i_tok_end]; text_proj [n_proj, d] F32,
		 * transposed before mul_mat => feat[j] = sum_i text_proj[j + n_proj*i] ... see below */
		const OParam *p = par(C, "text_proj", ORC_F32, d, d, 1, 1);
		OT *f = ot_new(d, 1, 1, 1);
		const float *xe = x->d + (int64_t)i_tok_end*d;
		/* p has ne = [n_proj, d]; transpose+cont gives pt[ne0=d, ne1=n_proj] with pt[i + d*j] = p[j + n_proj*i];
		 * mul_mat(pt, x): out[j] = sum_i pt[i + d*j] * x[i] = sum_i p[j + n_proj*i] * x[i] */
		for (int j=0;j<d;++j) { double s=0; for (int i=0;i<d;++i) s += (double)p->d[j + (int64_t)d*i] * xe[i]; f->d[j] = (float)s; }
		ot_free(x);
		return f;
	}
	return x;
}
