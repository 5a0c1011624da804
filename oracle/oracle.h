/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU (plain C + OpenMP) restatement of the reference's Stable Diffusion hot path
 * (aagdev/mlimgsynth @ 2025-06-14): the op graph that src/unet.c, src/vae.c,
 * src/tae.c, src/clip.c build through src/mlblock_nn.c / src/ggml_extend.c, and the
 * host loop of src/sampling.c, src/solvers.c, src/ccommon/rng_philox.c.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library, and only as the checker / reported baseline.  The product
 * (libmlimgsynth_amd.so) never links it.
 *
 * PARITY STATUS
 *   pinned   : Philox RNG (reference KAT src/test_rng.c:11-24, and oracle/_ref built
 *              from src/ccommon/rng_philox.c), sigma table end points
 *              (src/unet.c:34-35), 20-step uniform schedule (values
 *              obtained from the reference's own sampling.c/unet.c).
 *   UNPINNED : op arithmetic.  All tensor arithmetic of the reference lives in ggml,
 *              which is not vendored, not version-pinned and absent here (reference
 *              Makefile:18-31, .gitignore:6).  Op semantics are restated from the
 *              reference call sites plus ggml's published behaviour
 *              and cross-checked against torch CPU fp32 in tests/ — "parity unpinned".
 *
 * Conventions follow the reference: tensors are fp32, shape ne[0..3] with ne[0]
 * fastest; activations [W,H,C,N] (= NCHW memory), sequences [d,T,N]
 * (src/localtensor.h:16-20).  The oracle is batch-1 like the reference
 * (src/mlimgsynth.c:1640-1641).
 */
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ tensors */
struct OT;
typedef struct OT {
	int64_t ne[4];
	float *d;
} OT;

ORACLE_API OT*  ot_new(int64_t n0, int64_t n1, int64_t n2, int64_t n3);
ORACLE_API OT*  ot_from(const float* src, int64_t n0, int64_t n1, int64_t n2, int64_t n3);
ORACLE_API void ot_free(OT* t);
ORACLE_API int64_t ot_nel(const OT* t);

/* round-to-nearest-even through IEEE binary16 and back (ggml_fp32_to_fp16_row) */
ORACLE_API void orc_round_f16(float* x, int64_t n);
ORACLE_API void orc_set_threads(int n);
/* test switch (default 1): 0 disables the F16 rounding of conv/linear activation operands (see o_ops.c) */
ORACLE_API void orc_set_act_rounding(int on);
ORACLE_API void orc_set_trace(void (*cb)(const char* path, const struct OT* t));
ORACLE_API void orc_exp_sub(float* x, int64_t n, float mx);   /* x[j] = exp(x[j] - mx), the 8-lane polynomial of orc_attention's softmax (test hook) */
ORACLE_API void orc_set_linear_wtype(int type);   /* ORC_F16 (default) | ORC_F32: linear weights as an fp32 checkpoint gives them (src/mlimgsynth.c:1235-1236) */
ORACLE_API int orc_get_linear_wtype(void);
ORACLE_API int  orc_get_threads(void);
/* SGEMM micro-kernel: 2 = AVX2 6x16, 5 = AVX-512 12x32 (only where the CPU has it), anything else = choose at first use; both give the same bits */
ORACLE_API void orc_set_isa(int isa);
ORACLE_API int  orc_get_isa(void);
ORACLE_API void orc_prof_dump(int reset);
ORACLE_API void orc_bcache_drop(void);         /* the block cache (op outputs / scratch of >= 256 KB are kept for reuse) back to the system */      /* diagnostics: wall-clock seconds per op family since the last reset, to stderr */

/* C[M][N] = sum_k A[M][K] * B[N][K]   (ggml_mul_mat semantics: both K-contiguous) */
ORACLE_API void orc_sgemm_nt(int64_t M, int64_t N, int64_t K,
	const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc);

/* ------------------------------------------------------------------ params */
enum { ORC_F32 = 0, ORC_F16 = 1 };  /* same numbering as ggml_type (mlimgsynth.h:336-339) */

typedef struct OParam {
	char *name;
	int type;          /* ORC_F32 / ORC_F16: F16 params hold f16-representable values */
	int64_t ne[4];
	float *d;
} OParam;

typedef struct OParams OParams;

ORACLE_API OParams* orc_params_new(uint64_t synth_seed);   /* synth_seed: see oracle/o_params.c */
ORACLE_API void     orc_params_free(OParams* P);
/* explicit value (copied; rounded through f16 if type==ORC_F16) */
ORACLE_API int      orc_params_set(OParams* P, const char* name, int type,
	int64_t n0, int64_t n1, int64_t n2, int64_t n3, const float* data);
/* lookup; if absent, synthesised deterministically from (seed, name, shape) */
ORACLE_API const OParam* orc_params_get(OParams* P, const char* name, int type,
	int64_t n0, int64_t n1, int64_t n2, int64_t n3);
ORACLE_API int      orc_params_count(const OParams* P);
ORACLE_API const OParam* orc_params_at(const OParams* P, int i);
/* the synthetic generator itself (also restated in the product; tests compare them) */
ORACLE_API void orc_synth_fill(float* out, int64_t n, uint64_t seed, const char* name,
	float offset, float scale, int round_f16);
ORACLE_API void orc_synth_rule(const char* name, int type, const int64_t ne[4],
	float* offset, float* scale);

/* ------------------------------------------------------------------ ops (src/mlblock_nn.c, src/ggml_extend.c) */
ORACLE_API OT* orc_linear(const OT* x, const OParam* w, const OParam* b);
ORACLE_API OT* orc_conv2d(const OT* x, const OParam* w, const OParam* b, int s, int p);
ORACLE_API OT* orc_group_norm(const OT* x, int n_grp, float eps, const OParam* w, const OParam* b);
ORACLE_API OT* orc_layer_norm(const OT* x, float eps, const OParam* w, const OParam* b);
/* q [d_head*n_head... see o_ops.c */
ORACLE_API OT* orc_attention(const OT* q, const OT* k, const OT* v, int n_head, int causal);
ORACLE_API void orc_silu(OT* x);
/* 1: GELU / quick-GELU through ggml-CPU's F16 lookup tables (input and output rounded to binary16; restated from ggml's published source, ggml is absent: unpinned);
 * 0 (default): the exact fp32 formulas */
ORACLE_API void orc_set_ggml_f16_tables(int on);
ORACLE_API int  orc_get_ggml_f16_tables(void);
ORACLE_API void orc_gelu(OT* x);
ORACLE_API void orc_gelu_quick(OT* x);
ORACLE_API void orc_relu(OT* x);
ORACLE_API OT* orc_upscale2(const OT* x);
ORACLE_API OT* orc_pad_end(const OT* x, int p0, int p1);
ORACLE_API OT* orc_concat_ch(const OT* a, const OT* b);
ORACLE_API OT* orc_nchw_to_tokens(const OT* x);              /* [W,H,C,1] -> [C,W*H,1] */
ORACLE_API OT* orc_tokens_to_nchw(const OT* x, int w, int h); /* [C,T,1] -> [W,H,C,1] */
ORACLE_API void orc_timestep_embedding(const float* t, int n_t, int dim, float max_period, float* out);

/* ------------------------------------------------------------------ models */
typedef struct {
	int n_ch_in, n_ch_out, n_res_blk;
	int attn_res[4];
	int ch_mult[5];
	int transf_depth[5];
	int n_te, n_head, d_head, n_ctx, n_ch, ch_adm_in;
	int clip_norm, cond_label, uncond_empty_zero, vparam;
	int n_step_train;
	float sigma_min, sigma_max;
} OrcUnetParams;   /* mirrors UnetParams, src/unet.h:10-33 */

ORACLE_API void orc_unet_params_get(const char* model, OrcUnetParams* out); /* "sd1","sd2","sdxl","tiny","tinyxl" */

typedef struct {
	int ch_x, ch_z, ch, n_res, n_res_blk;
	int ch_mult[5];
	int d_embed, f_down;
	float scale_factor;
} OrcVaeParams;    /* mirrors VaeParams, src/vae.h:10-20 */

ORACLE_API void orc_vae_params_get(const char* model, OrcVaeParams* out);   /* "sd1","sdxl","tiny" */

typedef struct { int ch_x, ch_inner, ch_z, n_blk; } OrcTaeParams; /* src/tae.h */

typedef struct {
	int n_vocab, n_token, d_embed, n_interm, n_head, n_layer;
	int tok_start, tok_end, tok_pad;
} OrcClipParams;   /* mirrors ClipParams, src/clip.h */

ORACLE_API void orc_clip_params_get(const char* model, OrcClipParams* out); /* "vit_l","vit_h","vit_bigg","tiny" */

/* raw UNet graph: x [lw,lh,4,1] (already scaled by c_in), t (timestep), ctx [n_ctx,77,1],
 * label [ch_adm_in] or NULL -> out [lw,lh,4,1]           (mlb_unet_denoise, src/unet.c:263-281) */
ORACLE_API OT* orc_unet_graph(OParams* P, const char* prefix, const OrcUnetParams* U,
	const OT* x, float t, const OT* ctx, const OT* label);
/* unet_denoise_run (src/unet.c:460-498): sigma->t, c_in scaling, graph, v-param */
ORACLE_API OT* orc_unet_denoise_run(OParams* P, const char* prefix, const OrcUnetParams* U,
	const OT* x, const OT* cond, const OT* label, float sigma);
/* sdvae_decode (src/vae.c:318-411) without tiling, incl. (x+1)/2 post (src/vae.h:43-47) */
ORACLE_API OT* orc_vae_decode(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* latent);
/* sdtae_decode (src/tae.c:117-136) */
ORACLE_API OT* orc_tae_decode(OParams* P, const char* prefix, const OT* latent);
/* clip_text_encode (src/clip.c:439-488): tokens already padded to n_token.
 * returns embed [d,77,1]; if feat!=NULL also computes pooled feature [d] at i_tok_end (needs all layers+norm) */
ORACLE_API OT* orc_clip_text_encode(OParams* P, const char* prefix, const OrcClipParams* C,
	const int32_t* tokens, int clip_skip, int norm, int want_feat, int i_tok_end);

/* ------------------------------------------------------------------ sampling (src/sampling.c, src/solvers.c, src/unet.c:283-334) */
ORACLE_API void  orc_log_sigmas(float* out1000);
ORACLE_API float orc_sigma_to_t(float sigma);
ORACLE_API float orc_t_to_sigma(float t);
/* dnsamp_init uniform/karras schedule: writes n_step+1 sigmas, returns n_step */
ORACLE_API int   orc_schedule(int n_step_req, int sched, float f_t_ini, float f_t_end, float* sigmas);
ORACLE_API void  orc_ancestral(float s1, float s2, float eta, float* s_down, float* s_up);

typedef struct { uint64_t seed; uint32_t offset; } OrcRng;  /* RngPhilox, src/ccommon/rng_philox.h */
ORACLE_API void orc_rng_randn(OrcRng* S, unsigned n, float* out);
ORACLE_API void orc_philox_raw(uint64_t seed, uint32_t offset, unsigned n, uint32_t* out2n);

/* full txt2img denoise of ONE image: latent zeros -> final latent (mlis_generate loop,
 * src/mlimgsynth.c:1669-1753).  uncond/unlabel may be NULL when cfg_scale<=1.
 * nfe_limit>0 stops after that many UNet evaluations (for bounded CPU-baseline timing);
 * returns number of UNet evaluations done. */
ORACLE_API int orc_generate_latent(OParams* P, const char* prefix, const OrcUnetParams* U,
	int lw, int lh, const OT* cond, const OT* label, const OT* uncond, const OT* unlabel,
	float cfg_scale, int n_step, float s_ancestral, uint64_t seed, int nfe_limit,
	float* latent_out, double* t_unet_seconds);

/* ---- "next" rows: image encoders, latent sample, mask downsize, general sampler (all solvers, Karras, s_noise, img2img,
 * in-painting).  method: 1 euler, 2 heun, 3 taylor3, 4 dpmpp2m, 5 dpmpp2s (MLIS_Method); sched: 1 uniform, 2 karras */
ORACLE_API OT* orc_vae_encode_moments(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* img);
ORACLE_API OT* orc_vae_decode_tiled(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* latent, int tile_px);
ORACLE_API OT* orc_vae_encode_moments_tiled(OParams* P, const char* prefix, const OrcVaeParams* V, const OT* img, int tile_px);
ORACLE_API OT* orc_latent_sample(const OT* moments, const OrcVaeParams* V, const float* rnd);
ORACLE_API OT* orc_tae_encode(OParams* P, const char* prefix, const OT* img);
ORACLE_API void orc_mask_downsize(const float* mask, int w, int h, int f, float* lmask);
typedef struct { int method, sched, n_step; float cfg_scale, s_ancestral, s_noise, f_t_ini, f_t_end; } OrcSampleOpts;
ORACLE_API int orc_sample_ex(OParams* P, const char* prefix, const OrcUnetParams* U, int lw, int lh,
	const OT* cond, const OT* label, const OT* uncond, const OT* unlabel, const OrcSampleOpts* O,
	uint64_t seed, uint32_t rng_offset, const float* init_latent, const float* lmask, float* latent_out);

#ifdef __cplusplus
}
#endif
