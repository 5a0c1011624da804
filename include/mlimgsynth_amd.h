/* mlimgsynth_amd.h — model graphs, sampler and the generation driver of the MI355X engine.
 *
 * Mirrors, for the hot path only, the reference's model/sampler interfaces:
 *   UnetParams / unet_*      src/unet.h:10-62        VaeParams / sdvae_decode   src/vae.h:10-53
 *   SdTaeParams / sdtae_*    src/tae.h               ClipParams / clip_text_*   src/clip.h
 *   DenoiseSampler           src/sampling.h:17-46    Solver (Euler)             src/solvers.h:65-77
 *   RngPhilox                src/ccommon/rng_philox.h:10-22
 *   mlis_generate slice      include/mlimgsynth.h:414-558, src/mlimgsynth.c:1634-1773
 * Differences that are the point of the exercise: every graph takes a batch dimension N
 * (the reference rejects n_batch > 1, src/mlimgsynth.c:1640), cond and uncond evaluations
 * of classifier-free guidance run as ONE batch-2B UNet evaluation, and the latent, the
 * Euler(-ancestral) update and the CFG mix stay on the device between steps.
 * Return convention: >=1 ok, <0 error (reference TRY convention), text in mlsd_last_error().
 */
#pragma once
#include "mlblock_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- UNet (src/unet.h) */
typedef struct {
	int n_ch_in, n_ch_out, n_res_blk;
	int attn_res[4];
	int ch_mult[5];
	int transf_depth[5];
	int n_te, n_head, d_head, n_ctx, n_ch, ch_adm_in;
	int clip_norm, cond_label, uncond_empty_zero, vparam;
	int n_step_train;
	float sigma_min, sigma_max;
} UnetParams;

/* "sd1" | "sd2" | "sdxl" (g_unet_sd1/sd2/sdxl, src/unet.c:21-83) | "tiny" | "tinyxl" (test configs) */
int unet_params_get(const char* model, UnetParams* out);

MLTensor* mlb_unet_denoise(MLCtx* C, MLTensor* x, MLTensor* time, MLTensor* c, MLTensor* label, const UnetParams* P);

void  unet_params_init(void);
float unet_sigma_to_t(const UnetParams* P, float sigma);
float unet_t_to_sigma(const UnetParams* P, float t);

typedef struct {
	MLCtx* ctx;
	const UnetParams* par;
	unsigned nfe;
	int lw, lh, n_batch;           /* graph batch N (= 2*images with CFG) */
	MLTensor *t_x, *t_t, *t_c, *t_l, *t_out;
} UnetState;

/* (LocalTensor, the host tensor of the reference's boundary == MLIS_Tensor of include/mlimgsynth.h:409-413, and its ltensor_*
 * calls are declared in mlblock_amd.h) */

/* the reference's own entry points, same signatures (src/unet.h:55-62): batch 1, graph built at init, weights loaded afterwards
 * (mlctx_params_synth / mlctx_tstore_load on C).  `split` (--unet-split weight streaming, src/unet.c:390-458) is accepted and
 * ignored: weights are resident.  x [lw,lh,4,1], cond [n_ctx,77,1,1], label [adm,1,1,1] or NULL -> dx like x (resized). */
int unet_denoise_init(UnetState* S, MLCtx* C, const UnetParams* P, unsigned lw, unsigned lh, bool split);
int unet_denoise_run(UnetState* S, const LocalTensor* x, const LocalTensor* cond, const LocalTensor* label, float sigma, LocalTensor* dx);

/* builds the batch-N graph on C (prefix "unet"); weights are loaded afterwards with
 * mlctx_params_synth / mlctx_param_set */
int unet_denoise_init_n(UnetState* S, MLCtx* C, const UnetParams* P, unsigned lw, unsigned lh, unsigned n_batch);
/* second half of the init: records the graph and calls mlctx_prep (split so that the x input can first be
 * bound to a device-resident latent with mlctx_input_bind) */
int unet_denoise_build(UnetState* S);
/* host-boundary evaluation (tests, drop-in for src/unet.c:460-498 with a batch): x [N][4][lh][lw] NCHW,
 * cond [N][77][n_ctx], label [N][adm] or NULL, sigma[N] -> dx like x.  Applies c_in, sigma->t, v-param. */
int unet_denoise_run_n(UnetState* S, const float* x, const float* cond, const float* label,
	const float* sigma, float* dx);

/* ---------------------------------------------------------------- VAE / TAE decoders */
typedef struct {
	int ch_x, ch_z, ch, n_res, n_res_blk;
	int ch_mult[5];
	int d_embed, f_down;
	float scale_factor;
} VaeParams;
int vae_params_get(const char* model, VaeParams* out);   /* "sd1" | "sdxl" | "tiny" */
MLTensor* mlb_sdvae_decoder(MLCtx* C, MLTensor* x, const VaeParams* P);
/* builds the decode graph for n images of lw x lh latents on C (prefix "vae"); result [8lw,8lh,3,n] */
int sdvae_decode_init(MLCtx* C, const VaeParams* P, unsigned lw, unsigned lh, unsigned n_batch, MLTensor** t_latent);
/* host-boundary decode incl. the (x+1)/2 post (src/vae.h:43-47): latent NCHW [n][4][lh][lw] -> img [n][3][8lh][8lw] */
int sdvae_decode_run(MLCtx* C, MLTensor* t_latent, const float* latent, float* img);
/* graph build + prep only (the driver binds the resident latent and calls mlctx_compute itself) */
int sdvae_decode_build(MLCtx* C, const VaeParams* P, MLTensor* t_latent);

/* encoder (img2img / in-painting source; src/vae.c:76-128,231-316): image [n][3][h][w] in [0,1] -> moments
 * [n][2*ch_z][h/8][w/8] = (mean | logvar); the latent sample is a separate step (mlsd_latent_sample) */
MLTensor* mlb_sdvae_encoder(MLCtx* C, MLTensor* x, const VaeParams* P);
int sdvae_encode_init(MLCtx* C, const VaeParams* P, unsigned w, unsigned h, unsigned n_batch, MLTensor** t_img);
int sdvae_encode_build(MLCtx* C, const VaeParams* P, MLTensor* t_img);
int sdvae_encode_run(MLCtx* C, MLTensor* t_img, const float* img, float* moments);

typedef struct { int ch_x, ch_inner, ch_z, n_blk; } SdTaeParams;
MLTensor* mlb_sdtae_decoder(MLCtx* C, MLTensor* x, const SdTaeParams* P);
int sdtae_decode_init(MLCtx* C, unsigned lw, unsigned lh, unsigned n_batch, MLTensor** t_latent);
int sdtae_decode_run(MLCtx* C, MLTensor* t_latent, const float* latent, float* img);
int sdtae_decode_build(MLCtx* C, MLTensor* t_latent);
MLTensor* mlb_sdtae_encoder(MLCtx* C, MLTensor* x, const SdTaeParams* P);       /* src/tae.c:43-63 */
int sdtae_encode_init(MLCtx* C, unsigned w, unsigned h, unsigned n_batch, MLTensor** t_img);
int sdtae_encode_build(MLCtx* C, MLTensor* t_img);
int sdtae_encode_run(MLCtx* C, MLTensor* t_img, const float* img, float* latent);

/* ---------------------------------------------------------------- CLIP text encoder (src/clip.h) */
typedef struct {
	int n_vocab, n_token, d_embed, n_interm, n_head, n_layer;
	int tok_start, tok_end, tok_pad;
} ClipParams;
int clip_params_get(const char* model, ClipParams* out);   /* "vit_l" | "vit_h" | "vit_bigg" | "tiny" */
/* clip_text_encode (src/clip.c:439-488) for a batch of n prompts: toks [n][n_tok] (without BOS/EOS/PAD,
 * all prompts padded by the caller to the same n_tok), embed out [n][77][d] or NULL, feat out [n][d] or NULL */
int clip_text_encode(MLCtx* C, const ClipParams* P, const char* tprefix, unsigned n_prompt, unsigned n_tok,
	const int32_t* toks, float* embed, float* feat, int clip_skip, bool norm, uint64_t synth_seed);

/* resident variant: graph + weights stay on the device between prompts (the reference rebuilds the graph
 * and re-uploads the weights for every clip_text_encode call, src/clip.c:458-486) */
typedef struct {
	MLCtx* C;
	ClipParams P;
	MLTensor *t_tokens, *t_embed;
	unsigned n_prompt;
	int want_feat;
	char prefix[16];
	float* text_proj_host;
	MLTensor* t_tap;          /* init_ex with tap_skip > 0: copy of the hidden state clip_skip = tap_skip layers from the end (no norm) */
	void* tap_dev;
} ClipEncoder;
int clip_encoder_init(ClipEncoder* E, MLCtx* C, const ClipParams* P, const char* tprefix, unsigned n_prompt,
	int clip_skip, bool norm, bool want_feat);         /* then load weights: mlctx_params_synth / mlctx_param_set */
int clip_encoder_run(ClipEncoder* E, unsigned n_tok, const int32_t* toks, float* embed, float* feat);
/* one pass for both things SDXL wants from open_clip bigG (src/mlimgsynth.c:1517-1543 runs the tower twice): the whole stack
 * with final norm (pooled feature) AND the un-normed hidden state `tap_skip` layers from the end (the embedding), tapped by a
 * device copy; and prompts of different lengths in one run (prompt + negative prompt): n_tok[p], toks[p] for p < n_used
 * (<= the n_prompt the encoder was built for).  embed / tap out [n_used][77][d], feat out [n_used][d]; any may be NULL. */
int clip_encoder_init_ex(ClipEncoder* E, MLCtx* C, const ClipParams* P, const char* tprefix, unsigned n_prompt,
	int clip_skip, bool norm, bool want_feat, int tap_skip);
int clip_encoder_run_ex(ClipEncoder* E, unsigned n_used, const int* n_tok, const int32_t* const* toks, float* embed, float* tap, float* feat);
void clip_encoder_free(ClipEncoder* E);
/* SDXL vector conditioning (src/mlimgsynth.c:1542-1557): [pooled | emb(h,w) | emb(0,0) | emb(h,w)], 256 dims each */
int sdxl_label_build(const float* feat, int n_feat, int width, int height, float* label, int n_label);

/* ---- text conditioning on resident towers: mlis_text_cond_encode / mlis_clip_tokens_encode
 * (src/mlimgsynth.c:1423-1468,1501-1563).  model "sd1" | "sdxl" | "tiny" | "tinyxl"; tokens without BOS/EOS/padding.
 * cond [77][n_ctx]; label [n_label] (SDXL: pooled bigG feature || size embeddings; NULL for SD1). */
typedef struct MLIS_AmdTextCond MLIS_AmdTextCond;
typedef struct MLIS_AmdCtx MLIS_AmdCtx;          /* the generation driver, below */
MLIS_AmdTextCond* mlis_amd_textcond_create(const char* model, int width, int height, uint64_t weight_seed, void* stream);
/* clip_skip 0 = model default; defer_weights != 0: the towers are built without weights, the caller loads them into
 * mlis_amd_textcond_ctx(T, i) for i < mlis_amd_textcond_n_towers(T) (mlctx_tstore_load / mlctx_param_set) */
MLIS_AmdTextCond* mlis_amd_textcond_create_ex(const char* model, int width, int height, uint64_t weight_seed, void* stream,
	int clip_skip, int defer_weights);
int mlis_amd_textcond_n_towers(const MLIS_AmdTextCond* T);
MLCtx* mlis_amd_textcond_ctx(MLIS_AmdTextCond* T, int i);
int mlis_amd_textcond_set_size(MLIS_AmdTextCond* T, int width, int height);     /* SDXL size embeddings of the label */
/* with per-token weights (prompt emphasis, src/mlimgsynth.c:1457-1463); weights NULL = all 1 */
int mlis_amd_textcond_encode_w(MLIS_AmdTextCond* T, const int32_t* toks, const float* weights, int n_tok, float* cond, float* label);
int mlis_amd_textcond_encode_pair_w(MLIS_AmdTextCond* T, const int32_t* toks, const float* w, int n_tok,
	const int32_t* neg, const float* nw, int n_neg, float* cond, float* label, float* ncond, float* nlabel);   /* both prompts, one run per tower */
void mlis_amd_textcond_destroy(MLIS_AmdTextCond* T);
int mlis_amd_textcond_dims(const MLIS_AmdTextCond* T, int* n_ctx, int* n_label);
double mlis_amd_textcond_flops(const MLIS_AmdTextCond* T);
int mlis_amd_textcond_encode(MLIS_AmdTextCond* T, const int32_t* toks, int n_tok, float* cond, float* label);
/* prompt + negative prompt; an EMPTY negative prompt on SDXL zeroes ncond (uncond_empty_zero, :1702-1703) */
int mlis_amd_textcond_encode_pair(MLIS_AmdTextCond* T, const int32_t* toks, int n_tok, const int32_t* neg, int n_neg,
	float* cond, float* label, float* ncond, float* nlabel);

/* prompt + negative prompt encoded and written into an engine's conditioning inputs in one call (launchers; rank 0 of a multi-GPU job) */
const void* mlis_amd_cond_device(MLIS_AmdCtx* S, int what);   /* the plan's conditioning inputs (0: cond, 1: label), read-only */
int mlis_amd_textcond_apply(MLIS_AmdTextCond* T, MLIS_AmdCtx* E, const int32_t* toks, int n_tok, const int32_t* neg, int n_neg);

/* ---- CLIP BPE tokenizer (host).  Replaces clip_tokenize and helpers, src/clip.c:59-278 (public entry
 * mlis_text_tokenize, include/mlimgsynth.h); pinned by the 14 KATs of src/test_text_tokenize_clip.c:41-66.
 * The merge table is run-time data here (the reference compiles src/clip_merges.c.h in): id pairs in rank order
 * (token 512+i = merge i) or OpenAI's public bpe_simple_vocab_16e6.txt / merges.txt.  All functions return
 * counts (>= 0) or < 0 on error (mlsd_last_error()). */
typedef struct ClipTokenizer ClipTokenizer;
ClipTokenizer* clip_tokr_new(void);
void clip_tokr_free(ClipTokenizer* T);
int clip_tokr_set_merges(ClipTokenizer* T, const int32_t* pairs /* [n][2] */, int n);
int clip_tokr_load_merges_txt(ClipTokenizer* T, const char* path, int max_merges /* <= 0: CLIP's 48894 */);
int clip_tokr_n_merges(const ClipTokenizer* T);
int clip_tokr_n_vocab(const ClipTokenizer* T);              /* 512 + merges + start/end */
int clip_tokr_byte_to_token(int byte);                      /* clip_tokr_byte_to_token, src/clip.c:113-125 */
int clip_tokr_token_to_byte(int token);
/* text (UTF-8, len < 0: NUL-terminated) -> token ids without BOS/EOS/padding; returns the count */
int clip_tokenize(const ClipTokenizer* T, const char* text, int64_t len, int32_t* out, int max_out);
/* bytes a byte/merge token stands for; returns the byte count (may exceed max: nothing is written past it) */
int clip_token_decode(const ClipTokenizer* T, int32_t token, char* out, int max, int* end_of_word);

/* ---------------------------------------------------------------- checkpoint loading
 * tensor-name conversion: tnconv_sd, src/tensor_name_conv.c:274-324 (0 unused, 1 converted, 2 = open_clip fused in_proj) */
enum { TNCONV_R_UNUSED = 0, TNCONV_R_GOOD = 1, TNCONV_R_QKV_PROJ = 2 };
int tnconv_sd(const char* name, char* out, size_t out_size);
/* tensor index over an mmap'd safetensors file (src/ccompute/tensorstore_safet.c:152-205, tensorstore.c:184-323) */
typedef struct MLTStore MLTStore;
typedef struct { char* name; int dtype; int n_dim; int64_t shape[4]; /* shape[0] fastest */ size_t size; const void* data; } MLTSEntry;
/* convert_names != 0: names go through tnconv_sd, unused tensors are dropped and open_clip in_proj tensors are split into
 * q/k/v_proj views (tensor_callback_main / open_clip_attn_conv, src/mlimgsynth.c:989-1055) */
MLTStore* mlts_open(const char* path, int convert_names);              /* safetensors or GGUF v2/v3, detected by content */
MLTStore* mlts_open_safetensors(const char* path, int convert_names);  /* same (kept name) */
int       mlts_entry_to_f32(const MLTSEntry* e, float* out, int64_t n); /* any stored type (incl. GGUF block-quantised) -> fp32 */
void mlts_close(MLTStore* S);
/* LoRA (src/lora.c:9-138, tensor_callback_lora src/mlimgsynth.c:1068-1092): open a kohya-named LoRA file, merge it into the
 * model store: W += (scale | alpha/rank | 1) * mult * up.down for every "<X>.lora_down.weight"; returns the number of tensors patched */
MLTStore* mlts_open_lora(const char* path);
int mlts_lora_apply(MLTStore* model, const MLTStore* lora, float mult, int wtype);
int mlts_count(const MLTStore* S);
const MLTSEntry* mlts_at(const MLTStore* S, int i);
const MLTSEntry* mlts_find(const MLTStore* S, const char* name);
int mlts_stats(const MLTStore* S, int* n_unused, int* n_split);
/* mlis_model_identify (src/mlimgsynth.c:1206-1249): "sd1" | "sd2" | "sdxl" or NULL; *wtype = dtype of the probe tensor */
const char* mlts_model_identify(const MLTStore* S, int* wtype);
/* mlctx_tstore_load (src/mlblock.c:266-292): every parameter of the prepared plan by name; element count checked */
int mlctx_tstore_load(MLCtx* C, const MLTStore* S);

/* ---------------------------------------------------------------- prompt pre-processing (src/prompt_preproc.h:104-209)
 * text = the prompt with emphasis marks / options removed; chunks index into it; loras name into lora_names */
typedef struct { int begin, len; float w; } MLISPromptChunk;
typedef struct { int name_off, len; float w; } MLISPromptLora;
typedef struct {
	char* text; int n_text;
	MLISPromptChunk* chunks; int n_chunk;
	char* lora_names; int n_lora_chars;
	MLISPromptLora* loras; int n_lora;
} MLISPrompt;
int mlis_prompt_set_raw(MLISPrompt* P, const char* text);       /* prompt_text_set_raw */
int mlis_prompt_set_parse(MLISPrompt* P, const char* text);     /* prompt_text_set_parse; MLIS_E_PROMPT_PARSE (-5) on error */
void mlis_prompt_free(MLISPrompt* P);

/* ---------------------------------------------------------------- RNG / schedule / sampler */
typedef struct { uint64_t seed; uint32_t offset; } RngPhilox;   /* src/ccommon/rng_philox.h */
void rng_philox_randn(RngPhilox* S, unsigned n, float* out);
/* values [i0, i1) of the draw the generator stands at, without advancing it (the engine cuts a draw into ranges over host threads) */
void rng_philox_randn_range(const RngPhilox* S, unsigned i0, unsigned i1, float* out);

enum { DNSAMP_SCHED_UNIFORM = 1, DNSAMP_SCHED_KARRAS = 2 };      /* src/sampling.h:11-14 */
enum { SOLVER_METHOD_EULER = 1, SOLVER_METHOD_HEUN = 2, SOLVER_METHOD_TAYLOR3 = 3, SOLVER_METHOD_DPMPP2M = 4,
       SOLVER_METHOD_DPMPP2S = 5 };                              /* src/solvers.h:56-62 == MLIS_Method */

/* dnsamp_init schedule (src/sampling.c:28-96): fills sigmas[0..n_step], returns n_step */
int  dnsamp_schedule(const UnetParams* P, int n_step, int sched, float f_t_ini, float f_t_end, float* sigmas);
void dnsamp_ancestral(float s1, float s2, float eta, float* s_down, float* s_up);

/* ---------------------------------------------------------------- generation driver (mlis_generate slice) */
typedef struct {
	const char* model;       /* "sd1" | "sdxl" | "tiny" | "tinyxl" */
	int width, height;       /* pixels (multiple of 8) */
	int n_batch;             /* images generated together on this GPU */
	int n_step;              /* 20 */
	float cfg_scale;         /* 7 */
	float s_ancestral;       /* 1 = euler_a */
	int sched;               /* DNSAMP_SCHED_UNIFORM */
	int use_tae;             /* decode with TAESD instead of the KL-VAE */
	int use_hipgraph;        /* replay each UNet evaluation as one hipGraph launch */
	uint64_t weight_seed;    /* synthetic weights seed (1234) */
	/* sampler options of dnsamp_init (src/sampling.h:34-38); zero = the reference's defaults */
	int method;              /* SOLVER_METHOD_* (0 -> euler, src/sampling.c:33); 2-NFE solvers halve the step count (:47-49) */
	float s_noise;           /* stochastic sampling noise level (src/sampling.c:139-151) */
	float f_t_ini, f_t_end;  /* relative initial / final time: 0,0 -> 1,0 (txt2img); f_t_ini < 1 = img2img */
	int defer_weights;       /* 1: do not synthesise weights: the caller loads them (mlctx_param_set on the *_ctx handles) */
	int unet_split;          /* > 0: the UNet's weights are STREAMED (the reference's --unet-split / MLIS_OPT_UNET_SPLIT, src/unet.c:390-458): master copy in pinned host
	                          * memory, three to five device slabs (3 for SDXL) of `unet_split` MiB each (1 = the default 512 MiB: 1.5 GiB of slabs + 0.68 GB of resident step-invariant
	                          * weights instead of 4.8 GiB) filled under the launches; excludes use_hipgraph */
} MLIS_AmdConfig;

/* progress callback (MLIS_Callback, include/mlimgsynth.h:405): called after every COMPLETED step (the stream is
 * synchronised first); a negative return aborts the generation and is returned by mlis_amd_denoise/generate */
typedef int (*mlis_amd_progress_fn)(void* user, int step, int n_step, int nfe);

MLIS_AmdCtx* mlis_amd_create(const MLIS_AmdConfig* cfg, void* stream);
void mlis_amd_destroy(MLIS_AmdCtx* S);
/* conditioning for the whole batch (shared prompt, as generate.sh): cond/uncond [77][n_ctx] fp32 host,
 * label/unlabel [adm] or NULL */
int mlis_amd_set_cond(MLIS_AmdCtx* S, const float* cond, const float* label, const float* uncond, const float* unlabel);
/* device-resident variant used after an RCCL broadcast: pointers are DEVICE pointers of the same shapes */
int mlis_amd_set_cond_device(MLIS_AmdCtx* S, const void* cond, const void* label, const void* uncond, const void* unlabel);
/* runs the denoising loop for n_batch images with seeds[i] (offset 0 per image, src/generate.sh:56-59
 * semantics) and decodes; everything is enqueued on the stream, the call returns after the final sync.
 * latents_out [n][4][lh][lw] and/or images_out [n][3][h][w] (host, may be NULL) */
int mlis_amd_generate(MLIS_AmdCtx* S, const uint64_t* seeds, float* latents_out, float* images_out);
int mlis_amd_set_callback(MLIS_AmdCtx* S, mlis_amd_progress_fn fn, void* user);
/* change the sampler options between generations (plans and weights stay); 0 / negative values = the reference's defaults */
int mlis_amd_set_sampler(MLIS_AmdCtx* S, int n_step, int method, int sched, float cfg_scale, float s_ancestral, float s_noise,
	float f_t_ini, float f_t_end);
/* per-image Philox streams (seed_i, offset 0).  mlis_amd_denoise/generate with seeds == NULL continue the streams, like the
 * reference's never-reset g_rng (src/ccommon/rng_philox.c:50) */
int mlis_amd_seed(MLIS_AmdCtx* S, const uint64_t* seeds);
int mlis_amd_seed_ex(MLIS_AmdCtx* S, const uint64_t* seeds, uint32_t offset);    /* resume streams at a Philox offset */
uint32_t mlis_amd_rng_offset(const MLIS_AmdCtx* S);                           /* calls made so far on every stream */
/* img2img / in-painting inputs (MLIS_TUF_LATENT / MLIS_TUF_LMASK, src/mlimgsynth.c:1652-1687): initial latent host NCHW
 * [n][4][lh][lw] (consumed by the next denoise), latent mask host [lh][lw] (1 = keep the original; NULL clears) */
int mlis_amd_set_init_latent(MLIS_AmdCtx* S, const float* latent);
int mlis_amd_set_lmask(MLIS_AmdCtx* S, const float* lmask);
/* mlis_image_encode (src/mlimgsynth.c:1301-1330): images host NCHW [n][3][h][w] in [0,1] -> resident latent (VAE: sampled
 * when sample != 0 with one Philox call per image, else the mean; TAE: direct), marked as the next initial latent */
int mlis_amd_encode(MLIS_AmdCtx* S, const float* images, int sample);
MLCtx* mlis_amd_encoder_ctx(MLIS_AmdCtx* S);                                /* NULL before the first encode / prepare */
MLCtx* mlis_amd_encoder_prepare(MLIS_AmdCtx* S);
/* VAE tiling (MLIS_OPT_VAE_TILE, src/vae.c:245-300,333-391): tile size in pixels (rounded up to 64; 0 = off).  Decode / encode then
 * run tile by tile through tile-sized plans; *_tile_prepare build them (NULL when tiling does not apply: TAE, or one tile covers all) */
int mlis_amd_set_vae_tile(MLIS_AmdCtx* S, int tile_px);
MLCtx* mlis_amd_decoder_tile_prepare(MLIS_AmdCtx* S);
MLCtx* mlis_amd_encoder_tile_prepare(MLIS_AmdCtx* S);                            /* build the encoder plan now (weights: synth or caller-loaded) */
int mlis_amd_last_n_step(MLIS_AmdCtx* S);
/* passes (denoising loop, decode, encode) that were run AGAIN on the hand-off-free plan after an in-launch hand-off (stream-K slab, LayerNorm statistics) timed out --
 * the GPU was shared or the stream CU-masked.  The call that hit it still succeeds; results then come from plain tiles / separate LayerNorm launches. */
int mlis_amd_handoff_retries(const MLIS_AmdCtx* S);
/* pieces, for tests and for the multi-GPU driver */
int mlis_amd_denoise(MLIS_AmdCtx* S, const uint64_t* seeds);               /* latent stays on device */
int mlis_amd_decode(MLIS_AmdCtx* S);                                        /* image stays on device; asynchronous UNLESS the decoder plan hands data over inside launches
                                                                             * (stream-K tiles: the SDXL VAE has them): then the stream is drained here to check the hand-offs */
int mlis_amd_sync(MLIS_AmdCtx* S);                                          /* wait for the engine's stream */
void* mlis_amd_latent_device(MLIS_AmdCtx* S);                               /* fp32 NCHW [n][4][lh][lw] */
void* mlis_amd_image_device(MLIS_AmdCtx* S);                                /* fp32 NCHW [n][3][h][w] */
/* multi-GPU exchange steps over RCCL (mlsd_rccl_* communicator): conditioning broadcast from `root` into the plan's inputs;
 * all-gather of the final latents (what 0) or images (what 1) into recv_dev [world][per-rank bytes] */
int mlis_amd_bcast_cond(MLIS_AmdCtx* S, void* comm, int root);
int mlis_amd_gather_results(MLIS_AmdCtx* S, void* comm, int what, void* recv_dev);
int mlis_amd_info(MLIS_AmdCtx* S, double* unet_flops_per_eval, double* decode_flops, int* unet_ops, size_t* mem_params,
	size_t* mem_compute);
MLCtx* mlis_amd_unet_ctx(MLIS_AmdCtx* S);
MLCtx* mlis_amd_decoder_ctx(MLIS_AmdCtx* S);
/* time of the last mlis_amd_denoise spent in UNet evaluations, measured with HIP events on the stream (ms) */
float mlis_amd_last_unet_ms(MLIS_AmdCtx* S);
int mlis_amd_last_nfe(MLIS_AmdCtx* S);

#ifdef __cplusplus
}
#endif
