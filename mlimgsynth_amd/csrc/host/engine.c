/* Generation driver — the slice of the reference's mlis_generate (src/mlimgsynth.c:1634-1773) that drives the hot
 * path: [image encode] -> schedule -> initial noise -> n_step x [dxdt (CFG UNet evaluation) + solver update
 * (+ ancestral / stochastic noise, in-painting blend)] -> latent decode.  The loop is a re-creation of
 * dnsamp_init / dnsamp_step (src/sampling.c:28-185) and of the five solvers (src/solvers.c:82-296) whose vector
 * loops run on the device (csrc/hip/sampler.hip, same fp32/fp64 operation order) while every scalar stays on the
 * host exactly as in the reference.  MI355X-first differences:
 *   - n_batch images per GPU; cond and uncond of all images form ONE batch-2B UNet evaluation
 *     (the reference runs two sequential batch-1 evaluations per dxdt, mlimgsynth.c:1578-1582);
 *   - the latent never leaves the device: c_in scaling happens in the UNet's input conversion, the CFG mix and
 *     the solver updates are elementwise kernels with by-value scalars; nothing in the loop synchronises
 *     (unless a progress callback is installed);
 *   - the Philox/Box-Muller noise is generated on the host (bit-exact integer stream, fp64 Box-Muller like
 *     src/ccommon/rng_philox.c) WHILE the GPU runs the UNet evaluation enqueued just before, then copied
 *     asynchronously; one independent stream per image (seed_i, offset 0: generate.sh:56-59 semantics);
 *   - weights stay resident in HBM (288 GB) for the lifetime of the context instead of being re-uploaded per
 *     generate (mlblock.c:266-292) or per half-graph (--unet-split, unet.c:390-458).
 */
#define _GNU_SOURCE       /* sched_getaffinity, CPU_COUNT */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <unistd.h>

int unet_denoise_build(UnetState* S);

#define MAX_STEPS 1001     /* the public option STEPS accepts 0..1000 like the reference (mlimgsynth_options_set.c.h); buffers that follow the step count are sized in ensure_steps */
#define MAX_BATCH 64
#define MAX_DRAWS (2 * MAX_STEPS + 2)
#define N_TMP 2

struct MLIS_AmdCtx {
	MLIS_AmdConfig c;
	char model[16];
	void* stream;
	UnetParams unet_p;
	VaeParams vae_p;
	int lw, lh, hw, B, N;
	MLCtx *unet_ctx, *dec_ctx, *enc_ctx;
	UnetState unet;
	MLTensor *t_lat_dec, *t_img_enc;
	/* device state */
	float *d_x;                 /* latent [B][4][hw] NCHW fp32 */
	float *d_xsave;             /* copy of a caller-given initial latent while the denoising loop may have to be re-run (hand-off retry) */
	int n_handoff_retries;      /* denoise / decode / encode passes re-run on the hand-off-free plan after a timed-out in-launch hand-off */
	float *d_xin;               /* what the UNet plan reads (1 MB d2d copy per evaluation; keeps the plan independent of the solver) */
	float *d_dx;                /* dxdt */
	float *d_tmp[N_TMP];        /* solver state vectors (solver_tmp_get, solvers.c:54-76) */
	float *d_x0, *d_lmask;      /* in-painting: original latent, latent mask [hw] */
	float *d_img;               /* image  [B][3][H][W] fp32 */
	float *d_img_in;            /* encoder input image (NCHW fp32 [B][3][H][W], [0,1]) */
	float *d_cin;               /* [B] current c_in (read by the UNet input conversion) */
	float *d_noise;             /* [MAX draws][B][4*hw] */
	float *h_noise;             /* pinned, same shape */
	void *up_stream;            /* the draws are uploaded on a stream of their own (a 1 MB copy queued on the compute stream waits behind the copy engine's current job:
	                             * with streamed weights that is a 500 MB upload, 9 ms of an 80 ms evaluation); ev_draw[k] orders draw k before its consumer */
	void **ev_draw; int n_ev_draw;
	int32_t *d_nan;
	void *ev[2 * MAX_STEPS][2];
	int n_ev;
	float last_unet_ms;
	int last_nfe, last_n_step;
	int cond_set, have_init_latent, have_lmask;
	int own_stream;
	RngPhilox rng[MAX_BATCH];
	int cap_steps;
	int n_draw_max, n_draw_gen, k_draw;  /* noise draws of the current denoise call: planned / generated so far / consumed so far */
	int i_eval;
	mlis_amd_progress_fn cb; void* cb_user;
	/* VAE tiling (MLIS_OPT_VAE_TILE; src/vae.c:245-300,333-391): tile-sized plans, built on first use */
	int vae_tile, tl_w, tl_h, ti_w, ti_h;             /* option (pixels); latent / image tile size of the decode and encode plans */
	MLCtx *dect_ctx, *enct_ctx;
	MLTensor *t_lat_dect, *t_img_enct;
	float *d_lat_tile, *d_img_tile, *d_imgin_tile, *d_mom_tile, *d_mom;
};

static int fail(const char* msg) { return mlsd_set_error(-1, "%s", msg); }

MLB_API void mlis_amd_destroy(MLIS_AmdCtx* S)
{
	if (!S) return;
	mlsd_stream_sync(S->stream);
	if (S->unet_ctx) mlctx_destroy(S->unet_ctx);
	if (S->dec_ctx) mlctx_destroy(S->dec_ctx);
	if (S->enc_ctx) mlctx_destroy(S->enc_ctx);
	if (S->dect_ctx) mlctx_destroy(S->dect_ctx);
	if (S->enct_ctx) mlctx_destroy(S->enct_ctx);
	mlsd_free(S->d_lat_tile); mlsd_free(S->d_img_tile); mlsd_free(S->d_imgin_tile); mlsd_free(S->d_mom_tile); mlsd_free(S->d_mom);
	mlsd_free(S->d_xin); mlsd_free(S->d_xsave);
	mlsd_free(S->d_x); mlsd_free(S->d_dx); mlsd_free(S->d_x0); mlsd_free(S->d_lmask); mlsd_free(S->d_img); mlsd_free(S->d_img_in);
	for (int i=0;i<N_TMP;++i) mlsd_free(S->d_tmp[i]);
	mlsd_free(S->d_cin); mlsd_free(S->d_noise); mlsd_free(S->d_nan);
	mlsd_host_free(S->h_noise);
	for (int i=0;i<S->n_ev;++i) { mlsd_event_destroy(S->ev[i][0]); mlsd_event_destroy(S->ev[i][1]); }
	for (int i=0;i<S->n_ev_draw;++i) mlsd_event_destroy(S->ev_draw[i]);
	free(S->ev_draw);
	if (S->up_stream) { mlsd_stream_sync(S->up_stream); mlsd_stream_destroy(S->up_stream); }
	if (S->own_stream) mlsd_stream_destroy(S->stream);
	free(S);
}

/* buffers whose size follows the step count: noise draws (<= 2 per step + 2) and events (<= 2 per step) */
static int ensure_steps(MLIS_AmdCtx* S, int ns)
{
	if (ns >= MAX_STEPS) return fail("too many steps (max 1000)");
	if (ns <= S->cap_steps) return 1;
	const size_t lat_elems = (size_t)S->B * 4 * S->hw;
	mlsd_stream_sync(S->stream);
	if (S->up_stream) mlsd_stream_sync(S->up_stream);
	mlsd_free(S->d_noise); mlsd_host_free(S->h_noise);
	S->d_noise = NULL; S->h_noise = NULL; S->cap_steps = 0;
	S->n_draw_max = 2 * ns + 2;
	if (mlsd_malloc((void**)&S->d_noise, (size_t)S->n_draw_max*lat_elems*4)) return -1;
	if (mlsd_host_alloc((void**)&S->h_noise, (size_t)S->n_draw_max*lat_elems*4)) return -1;
	for (int i=S->n_ev; i<2*ns; ++i) { if (mlsd_event_create(&S->ev[i][0]) || mlsd_event_create(&S->ev[i][1])) return -1; S->n_ev = i+1; }
	if (!S->up_stream && mlsd_stream_create(&S->up_stream)) return -1;
	{	void **e = (void**)realloc(S->ev_draw, sizeof(void*) * (size_t)S->n_draw_max);
		if (!e) return fail("out of memory");
		S->ev_draw = e;
		for (int i=S->n_ev_draw; i<S->n_draw_max; ++i) { if (mlsd_event_create(&S->ev_draw[i])) return -1; S->n_ev_draw = i+1; }
	}
	S->cap_steps = ns;
	return 1;
}

static int solver_nfe(int method)
{	/* SolverClass.n_fe, src/solvers.c:90-296 */
	return (method == SOLVER_METHOD_HEUN || method == SOLVER_METHOD_DPMPP2S) ? 2 : 1;
}

MLB_API MLIS_AmdCtx* mlis_amd_create(const MLIS_AmdConfig* cfg, void* stream)
{
	MLIS_AmdCtx *S = (MLIS_AmdCtx*)calloc(1, sizeof(*S));
	if (!S) { fail("out of memory"); return NULL; }
	S->c = *cfg; S->stream = stream;
	if (!stream) {   /* own non-default stream: stream capture (hipGraph) is not permitted on the NULL stream */
		if (mlsd_stream_create(&S->stream)) { free(S); return NULL; }
		S->own_stream = 1;
		stream = S->stream;
	}
	snprintf(S->model, sizeof(S->model), "%s", cfg->model ? cfg->model : "sd1");
	S->c.model = S->model;
	if (S->c.n_step < 1) S->c.n_step = 20;
	if (S->c.n_step >= MAX_STEPS) { fail("too many steps (max 1000)"); goto err; }
	if (S->c.n_batch < 1) S->c.n_batch = 1;
	if (S->c.n_batch > MAX_BATCH) { fail("n_batch > 64 not supported"); goto err; }
	if (!(S->c.cfg_scale > 0)) S->c.cfg_scale = 7;          /* default cfg 7, src/mlimgsynth.c:474 */
	if (!S->c.sched) S->c.sched = DNSAMP_SCHED_UNIFORM;
	if (S->c.method <= 0) S->c.method = SOLVER_METHOD_EULER;   /* sampling.c:33 */
	if (S->c.method > SOLVER_METHOD_DPMPP2S) { mlsd_set_error(-1, "invalid sampling method %d", S->c.method); goto err; }
	if (!(S->c.f_t_ini > 0)) S->c.f_t_ini = 1;
	if (!S->c.weight_seed) S->c.weight_seed = 1234;
	if (unet_params_get(S->model, &S->unet_p) < 0) goto err;
	if (vae_params_get(S->model, &S->vae_p) < 0) goto err;
	if (S->c.width % 8 || S->c.height % 8 || S->c.width < 8 || S->c.height < 8) { fail("image size must be a multiple of 8"); goto err; }
	S->lw = S->c.width / S->vae_p.f_down; S->lh = S->c.height / S->vae_p.f_down; S->hw = S->lw * S->lh;
	S->B = S->c.n_batch;
	S->N = S->c.cfg_scale > 1 ? 2*S->B : S->B;
	const int B = S->B, N = S->N, ns = S->c.n_step;
	const size_t lat_elems = (size_t)B * 4 * S->hw;

	if (mlsd_malloc((void**)&S->d_x, lat_elems*4)) goto err;
	if (mlsd_malloc((void**)&S->d_xin, lat_elems*4)) goto err;   /* the UNet plan's input: the evaluation point (x, or x1 of heun / dpmpp2s) is copied in */
	if (mlsd_malloc((void**)&S->d_dx, lat_elems*4)) goto err;
	for (int i=0;i<N_TMP;++i) if (mlsd_malloc((void**)&S->d_tmp[i], lat_elems*4)) goto err;
	if (mlsd_malloc((void**)&S->d_x0, lat_elems*4)) goto err;
	if (mlsd_malloc((void**)&S->d_lmask, (size_t)S->hw*4)) goto err;
	if (mlsd_malloc((void**)&S->d_img, (size_t)B*3*S->c.width*S->c.height*4)) goto err;
	if (mlsd_malloc((void**)&S->d_cin, B*4)) goto err;
	if (mlsd_malloc((void**)&S->d_nan, 256)) goto err;
	if (ensure_steps(S, ns) < 0) goto err;

	/* ---- UNet plan, x bound to the resident evaluation point (c_in scaling + cond/uncond duplication in the gather) */
	S->unet_ctx = mlctx_new(stream);
	if (S->c.unet_split > 0) {     /* --unet-split: the UNet's weights are streamed through three slabs (mlblock.c "weight streaming"); one evaluation = one pass over them */
		if (mlctx_set_weight_streaming(S->unet_ctx, S->c.unet_split > 1 ? (size_t)S->c.unet_split << 20 : 0) < 0) goto err;
	} else if (S->c.use_hipgraph) mlctx_set_flags(S->unet_ctx, MLB_F_HIPGRAPH);
	if (unet_denoise_init_n(&S->unet, S->unet_ctx, &S->unet_p, S->lw, S->lh, N) < 0) goto err;
	if (mlctx_input_bind(S->unet.t_x, S->d_xin, B, S->d_cin, 1.0f, 0) < 0) { fail("input bind failed"); goto err; }
	if (unet_denoise_build(&S->unet) < 0) goto err;
	if (!S->c.defer_weights && mlctx_params_synth(S->unet_ctx, S->c.weight_seed) < 0) goto err;

	/* ---- decoder plan, latent input bound to the same resident latent */
	S->dec_ctx = mlctx_new(stream);
	if (S->c.use_tae) {
		if (sdtae_decode_init(S->dec_ctx, S->lw, S->lh, B, &S->t_lat_dec) < 0) goto err;
		if (mlctx_input_bind(S->t_lat_dec, S->d_x, B, NULL, 1.0f, 1) < 0) goto err;
		if (sdtae_decode_build(S->dec_ctx, S->t_lat_dec) < 0) goto err;
	} else {
		if (sdvae_decode_init(S->dec_ctx, &S->vae_p, S->lw, S->lh, B, &S->t_lat_dec) < 0) goto err;
		if (mlctx_input_bind(S->t_lat_dec, S->d_x, B, NULL, 1 / S->vae_p.scale_factor, 0) < 0) goto err;
		if (sdvae_decode_build(S->dec_ctx, &S->vae_p, S->t_lat_dec) < 0) goto err;
	}
	if (!S->c.defer_weights && mlctx_params_synth(S->dec_ctx, S->c.weight_seed) < 0) goto err;
	return S;
err:
	mlis_amd_destroy(S);
	return NULL;
}

/* sampler options may change between generations without rebuilding the plans (the reference re-reads them in dnsamp_init on
 * every mlis_generate).  cfg_scale may change freely on its side of 1 (crossing it changes the UNet batch: new context). */
MLB_API int mlis_amd_set_sampler(MLIS_AmdCtx* S, int n_step, int method, int sched, float cfg_scale, float s_ancestral, float s_noise,
	float f_t_ini, float f_t_end)
{
	if (n_step < 1) n_step = 20;
	if (method <= 0) method = SOLVER_METHOD_EULER;
	if (method > SOLVER_METHOD_DPMPP2S) return mlsd_set_error(-4, "invalid sampling method %d", method);
	if (sched && sched != DNSAMP_SCHED_UNIFORM && sched != DNSAMP_SCHED_KARRAS) return mlsd_set_error(-4, "invalid sampling scheduler %d", sched);
	if (!(cfg_scale > 0)) cfg_scale = 7;
	if ((cfg_scale > 1) != (S->c.cfg_scale > 1)) return fail("cfg_scale crossing 1 changes the UNet batch: create a new context");
	if (ensure_steps(S, n_step) < 0) return -1;
	S->c.n_step = n_step; S->c.method = method; S->c.sched = sched ? sched : DNSAMP_SCHED_UNIFORM; S->c.cfg_scale = cfg_scale;
	S->c.s_ancestral = s_ancestral; S->c.s_noise = s_noise; S->c.f_t_ini = f_t_ini > 0 ? f_t_ini : 1; S->c.f_t_end = f_t_end;
	return 1;
}

MLB_API int mlis_amd_set_callback(MLIS_AmdCtx* S, mlis_amd_progress_fn fn, void* user)
{
	S->cb = fn; S->cb_user = user;
	return 1;
}

static int fill_cond(MLIS_AmdCtx* S, MLTensor* t, const void* a, const void* b, size_t per_bytes, int kind)
{	/* rows 0..B-1 <- a (cond), rows B..2B-1 <- b (uncond); the shared prompt is replicated per image */
	char *dst = (char*)mlctx_input_device_ptr(t);
	for (int n=0; n<S->N; ++n) {
		const void *src = n < S->B ? a : b;
		if (!src) return fail("missing (un)conditioning");
		if (mlsd_memcpy(dst + (size_t)n*per_bytes, src, per_bytes, kind, S->stream)) return -1;
	}
	return 1;
}

static int set_cond(MLIS_AmdCtx* S, const void* cond, const void* label, const void* uncond, const void* unlabel, int kind)
{
	const UnetParams *P = &S->unet_p;
	if (fill_cond(S, S->unet.t_c, cond, uncond, (size_t)77*P->n_ctx*4, kind) < 0) return -1;
	if (P->ch_adm_in && fill_cond(S, S->unet.t_l, label, unlabel, (size_t)P->ch_adm_in*4, kind) < 0) return -1;
	if (mlsd_stream_sync(S->stream)) return -1;
	S->cond_set = 1;
	return 1;
}

MLB_API int mlis_amd_set_cond(MLIS_AmdCtx* S, const float* cond, const float* label, const float* uncond, const float* unlabel)
{
	return set_cond(S, cond, label, uncond, unlabel, 0);
}

MLB_API int mlis_amd_set_cond_device(MLIS_AmdCtx* S, const void* cond, const void* label, const void* uncond, const void* unlabel)
{
	return set_cond(S, cond, label, uncond, unlabel, 2);
}

/* ------------------------------------------------------------------ multi-GPU (one process per GPU, images sharded over ranks)
 * rank `root` has set the conditioning (mlis_amd_set_cond*); every rank then calls this: the plan's conditioning inputs are
 * broadcast in place over RCCL (cond [N][77][n_ctx] + label [N][adm] fp32: 1.3 MB for SDXL), no host hop, no per-step collective */
MLB_API int mlis_amd_bcast_cond(MLIS_AmdCtx* S, void* comm, int root)
{
	const UnetParams *P = &S->unet_p;
	if (mlsd_rccl_bcast(comm, mlctx_input_device_ptr(S->unet.t_c), (size_t)S->N*77*P->n_ctx*4, root, S->stream)) return -1;
	if (P->ch_adm_in && mlsd_rccl_bcast(comm, mlctx_input_device_ptr(S->unet.t_l), (size_t)S->N*P->ch_adm_in*4, root, S->stream)) return -1;
	S->cond_set = 1;
	return 1;
}

/* all-gather of the final latents (what = 0: [B][4][lh][lw] fp32 per rank) or decoded images (what = 1: [B][3][H][W] fp32) into
 * `recv_dev` (world x the per-rank size, rank-major) on every rank; asynchronous on the engine's stream */
MLB_API int mlis_amd_gather_results(MLIS_AmdCtx* S, void* comm, int what, void* recv_dev)
{
	const size_t nb = what ? (size_t)S->B*3*S->c.width*S->c.height*4 : (size_t)S->B*4*S->hw*4;
	return mlsd_rccl_all_gather(comm, what ? (void*)S->d_img : (void*)S->d_x, recv_dev, nb, S->stream) ? -1 : 1;
}

/* read-only view of the conditioning the UNet plan will use (what = 0: cond [N][77][n_ctx] fp32, 1: label [N][adm]): tests */
MLB_API const void* mlis_amd_cond_device(MLIS_AmdCtx* S, int what)
{
	MLTensor *t = what ? S->unet.t_l : S->unet.t_c;
	return t ? t->in_stage : NULL;
}

MLB_API int mlis_amd_seed(MLIS_AmdCtx* S, const uint64_t* seeds)
{
	for (int b=0;b<S->B;++b) { S->rng[b].seed = seeds[b]; S->rng[b].offset = 0; }
	return 1;
}

MLB_API int mlis_amd_seed_ex(MLIS_AmdCtx* S, const uint64_t* seeds, uint32_t offset)
{
	for (int b=0;b<S->B;++b) { S->rng[b].seed = seeds[b]; S->rng[b].offset = offset; }
	return 1;
}
MLB_API uint32_t mlis_amd_rng_offset(const MLIS_AmdCtx* S) { return S->rng[0].offset; }

/* initial latent for img2img (MLIS_TUF_LATENT, src/mlimgsynth.c:1661-1666): host NCHW [B][4][lh][lw], or NULL to go back
 * to txt2img (zero latent) */
MLB_API int mlis_amd_set_init_latent(MLIS_AmdCtx* S, const float* latent)
{
	if (!latent) { S->have_init_latent = 0; return 1; }
	if (mlsd_memcpy(S->d_x, latent, (size_t)S->B*4*S->hw*4, 0, S->stream) || mlsd_stream_sync(S->stream)) return -1;
	S->have_init_latent = 1;
	return 1;
}

/* latent mask for in-painting (MLIS_TUF_LMASK): host [lh][lw], 1 = keep the original latent; NULL clears it */
MLB_API int mlis_amd_set_lmask(MLIS_AmdCtx* S, const float* lmask)
{
	if (!lmask) { S->have_lmask = 0; return 1; }
	if (mlsd_memcpy(S->d_lmask, lmask, (size_t)S->hw*4, 0, S->stream) || mlsd_stream_sync(S->stream)) return -1;
	S->have_lmask = 1;
	return 1;
}

/* ------------------------------------------------------------------ noise draws
 * Every dnsamp_noise_add (src/sampling.c:112-117) is one rng_randn call of the whole latent per image.  Which draws a
 * run makes is known from the schedule alone, so draw k is generated on the host as soon as the evaluation before it
 * has been enqueued (the GPU is busy for tens of ms) and uploaded asynchronously. */
/* One draw = rng_philox_randn of the whole latent per image (src/ccommon/rng_philox.c:23-51): value i of image b is a pure function of (seed_b, offset_b, i), so the
 * B x 4 x hw values are cut into ranges over host threads -- bit-identical by construction (round 5: the single-threaded draw of the FIRST noise, 4 x 65 536 Box-Muller values
 * in fp64 before any GPU work, was 12 ms of every SDXL batch-4 step). */
typedef struct { const RngPhilox* rng; unsigned i0, i1; float* out; } DrawJob;
enum { DRAW_MAXT = 16, DRAW_MIN_CHUNK = 8192 };      /* (a range per 8192 values at least, 16 threads at most) */

/* CPUs this process is GRANTED, not the ones the machine has: scheduler affinity and the cgroup CPU quota -- the GPU box shows 256 logical CPUs and grants 16;
 * on a pod with a quota of 1-4 a thread per online CPU would oversubscribe the very thread that enqueues the GPU work. */
static int granted_cpus(void)
{
	long n = sysconf(_SC_NPROCESSORS_ONLN);
	if (n < 1) n = 1;
	cpu_set_t set;
	if (!sched_getaffinity(0, sizeof(set), &set)) { const int a = CPU_COUNT(&set); if (a >= 1 && a < n) n = a; }
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");                      /* cgroup v2: "<quota|max> <period>" */
	if (f) {
		char q[32]; long per = 0;
		if (fscanf(f, "%31s %ld", q, &per) == 2 && per > 0 && q[0] != 'm') { const long c = atol(q) / per; if (c >= 1 && c < n) n = c; else if (c < 1) n = 1; }
		fclose(f);
	} else {
		long q = -1, per = 0;                                             /* cgroup v1 */
		if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) { if (fscanf(f, "%ld", &q) != 1) q = -1; fclose(f); }
		if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) { if (fscanf(f, "%ld", &per) != 1) per = 0; fclose(f); }
		if (q > 0 && per > 0) { const long c = q / per; if (c >= 1 && c < n) n = c; else if (c < 1) n = 1; }
	}
	return (int)n;
}

/* A small PERSISTENT pool (process-wide, started at the first multi-range draw, alive until exit): a draw used to create and join up to 15 threads each time.  The caller
 * publishes the job array, works on it itself, and returns when every range is done; a pool of zero workers (one granted CPU, or pthread_create failing) degrades to the
 * caller running every range.  One draw at a time (call_mu): engines of different threads share the workers. */
static struct {
	pthread_mutex_t call_mu, mu; pthread_cond_t cv_work, cv_done;
	const DrawJob *jobs; int n_jobs, next, done, n_workers, started;
} g_draw = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, 0, 0 };

static void* draw_pool_worker(void* arg)
{
	(void)arg;
	pthread_mutex_lock(&g_draw.mu);
	for (;;) {
		while (g_draw.next >= g_draw.n_jobs) pthread_cond_wait(&g_draw.cv_work, &g_draw.mu);
		const DrawJob *j = &g_draw.jobs[g_draw.next++];
		pthread_mutex_unlock(&g_draw.mu);
		rng_philox_randn_range(j->rng, j->i0, j->i1, j->out);
		pthread_mutex_lock(&g_draw.mu);
		if (++g_draw.done == g_draw.n_jobs) pthread_cond_signal(&g_draw.cv_done);
	}
	return NULL;
}

static void draw_run(const DrawJob* jobs, int nj)
{
	if (nj <= 1) { for (int j=0;j<nj;++j) rng_philox_randn_range(jobs[j].rng, jobs[j].i0, jobs[j].i1, jobs[j].out); return; }
	pthread_mutex_lock(&g_draw.call_mu);
	pthread_mutex_lock(&g_draw.mu);
	if (!g_draw.started) {
		g_draw.started = 1;
		int want = granted_cpus(); if (want > DRAW_MAXT) want = DRAW_MAXT;
		pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
		for (int k=1;k<want;++k) { pthread_t t; if (pthread_create(&t, &at, draw_pool_worker, NULL)) break; g_draw.n_workers++; }      /* (the caller is thread 0) */
		pthread_attr_destroy(&at);
	}
	g_draw.jobs = jobs; g_draw.n_jobs = nj; g_draw.next = 0; g_draw.done = 0;
	pthread_cond_broadcast(&g_draw.cv_work);
	while (g_draw.next < g_draw.n_jobs) {
		const DrawJob *j = &g_draw.jobs[g_draw.next++];
		pthread_mutex_unlock(&g_draw.mu);
		rng_philox_randn_range(j->rng, j->i0, j->i1, j->out);
		pthread_mutex_lock(&g_draw.mu);
		g_draw.done++;
	}
	while (g_draw.done < g_draw.n_jobs) pthread_cond_wait(&g_draw.cv_done, &g_draw.mu);
	g_draw.jobs = NULL; g_draw.n_jobs = 0; g_draw.next = 0; g_draw.done = 0;
	pthread_mutex_unlock(&g_draw.mu);
	pthread_mutex_unlock(&g_draw.call_mu);
}

static void draw_all(MLIS_AmdCtx* S, float* hn, size_t per)
{
	static int ncpu = 0;
	if (!ncpu) { const int n = granted_cpus(); ncpu = n > DRAW_MAXT ? DRAW_MAXT : n; }
	const size_t total = (size_t)S->B * per;
	int nt = (int)(total / DRAW_MIN_CHUNK); if (nt > ncpu) nt = ncpu; if (nt < 1) nt = 1;
	int per_img = (nt + S->B - 1) / S->B; if (per_img < 1) per_img = 1;       /* ranges never straddle images */
	DrawJob jobs[MAX_BATCH * DRAW_MAXT];
	int nj = 0;
	for (int b=0;b<S->B;++b) for (int q=0;q<per_img;++q) {
		const unsigned i0 = (unsigned)(per * q / per_img), i1 = (unsigned)(per * (q + 1) / per_img);
		jobs[nj++] = (DrawJob){ &S->rng[b], i0, i1, hn + (size_t)b*per + i0 };
	}
	draw_run(jobs, nj);
	for (int b=0;b<S->B;++b) S->rng[b].offset++;
}

static const float* noise_gen(MLIS_AmdCtx* S, int k)
{	/* generate and upload every draw up to k (upload stream; nothing is enqueued on the compute stream) */
	const size_t per = (size_t)4 * S->hw, lat_elems = (size_t)S->B * per;
	if (k >= S->n_draw_max) { fail("internal: too many noise draws"); return NULL; }
	for (; S->n_draw_gen <= k; S->n_draw_gen++) {
		float *hn = S->h_noise + (size_t)S->n_draw_gen*lat_elems, *dn = S->d_noise + (size_t)S->n_draw_gen*lat_elems;
		draw_all(S, hn, per);
		if (mlsd_memcpy(dn, hn, lat_elems*4, 0, S->up_stream) || mlsd_event_record(S->ev_draw[S->n_draw_gen], S->up_stream)) return NULL;
	}
	return S->d_noise + (size_t)k*lat_elems;
}
static const float* noise_draw(MLIS_AmdCtx* S, int k)
{	/* draw k for a consumer enqueued next on the compute stream: that stream waits for the draw's upload */
	const float *nz = noise_gen(S, k);
	if (nz && mlsd_stream_wait_event(S->stream, S->ev_draw[k])) return NULL;
	return nz;
}

/* mlis_denoise_dxdt + unet_denoise_run_n (src/mlimgsynth.c:1565-1587, src/unet.c:460-498): one batched evaluation at
 * x_eval (device), sigma -> raw UNet output stays in the plan's result tensor.  `prefetch_draws`: noise draws to generate
 * on the host while this evaluation runs. */
static int unet_eval(MLIS_AmdCtx* S, const float* x_eval, float sigma, int prefetch_upto)
{
	const UnetParams *P = &S->unet_p;
	const int B = S->B, N = S->N;
	void *st = S->stream;
	if (S->i_eval >= 2*S->cap_steps) return fail("internal: too many evaluations");
	const float t = unet_sigma_to_t(P, sigma);
	const float c_in = 1 / sqrt(sigma*sigma + 1);                /* unet.c:471 */
	float *d_t_in = (float*)mlctx_input_device_ptr(S->unet.t_t);
	/* the same timestep for every row and the same c_in for every image: written by a kernel from its arguments.  (As two 64-byte copies they queued behind the copy
	 * engine's current job -- with streamed weights a 500 MB upload: 10 ms gaps in front of 4 of 20 evaluations, MLSD_ENGINE_TRACE.) */
	if (mlsd_fill2_f32(d_t_in, N, t, S->d_cin, B, c_in, st)) return -1;
	if (x_eval != S->d_xin && mlsd_memcpy(S->d_xin, x_eval, (size_t)B*4*S->hw*4, 2, st)) return -1;
	mlsd_event_record(S->ev[S->i_eval][0], st);
	if (mlctx_compute(S->unet_ctx) < 0) return -1;               /* cond + uncond of all images: one evaluation */
	mlsd_event_record(S->ev[S->i_eval][1], st);
	S->i_eval++;
	S->last_nfe += N / B;
	int64_t ld = 0;
	const float *eps = mlctx_tensor_device_f32(S->unet_ctx, S->unet.t_out, &ld);
	if (mlsd_count_nonfinite(eps, (size_t)N*S->hw*ld, S->d_nan, st)) return -1;   /* ltensor_finite_check, unet.c:487 */
	if (prefetch_upto >= 0 && !noise_gen(S, prefetch_upto)) return -1;
	return 1;
}

/* dx = CFG mix (+ v-param rescale) of the evaluation just enqueued */
static int dxdt_finish(MLIS_AmdCtx* S, const float* x_eval, float sigma, float* dx)
{
	int64_t ld = 0;
	const float *eps = mlctx_tensor_device_f32(S->unet_ctx, S->unet.t_out, &ld);
	const float c_skip = sigma / (sigma*sigma + 1), c_out = 1 / sqrt(sigma*sigma + 1);   /* unet.c:491-492 */
	return mlsd_dxdt_cfg(eps, ld, x_eval, dx, S->B, 4, S->hw, S->c.cfg_scale, S->unet_p.vparam, c_out, c_skip, S->stream) ? -1 : 1;
}

static int denoise_body(MLIS_AmdCtx* S, const uint64_t* seeds, int* retry_wanted)
{
	if (!S->cond_set) return fail("mlis_amd_denoise: conditioning not set");
	const UnetParams *P = &S->unet_p;
	const int B = S->B, method = S->c.method;
	const int64_t nel = (int64_t)B * 4 * S->hw;
	void *st = S->stream;
	if (seeds) mlis_amd_seed(S, seeds);

	/* ---- dnsamp_init (src/sampling.c:28-96) */
	float sigmas[MAX_STEPS+1];
	int n_req = S->c.n_step;
	const int nfe_solver = solver_nfe(method);
	if (nfe_solver > 1) n_req = (n_req + nfe_solver - 1) / nfe_solver;        /* keep the NFE count (:47-49) */
	const int n_step = dnsamp_schedule(P, n_req, S->c.sched, S->c.f_t_ini, S->c.f_t_end, sigmas);
	if (n_step < 0 || n_step > S->c.n_step) return fail("schedule failed");
	float solver_t = sigmas[0];                                               /* :92 */
	unsigned i_step_solver = 0;
	float dt_prev = 0, h_last = 0;                                            /* taylor3 / dpmpp2m scalar state */
	S->n_draw_gen = 0; S->i_eval = 0; S->last_nfe = 0; S->last_n_step = n_step;
	S->k_draw = 0;
	if (mlsd_memset(S->d_nan, 0, 4, st)) return -1;
	if (!S->have_init_latent && mlsd_memset(S->d_x, 0, (size_t)nel*4, st)) return -1;   /* mlimgsynth.c:1669-1670 */
	if (method == SOLVER_METHOD_TAYLOR3 || method == SOLVER_METHOD_DPMPP2M) {
		/* solver tmp tensors start zeroed (ltensor_resize of an empty tensor; values unused until i_step >= 1) */
		if (mlsd_memset(S->d_tmp[0], 0, (size_t)nel*4, st) || mlsd_memset(S->d_tmp[1], 0, (size_t)nel*4, st)) return -1;
	}

	for (int s=0; s<n_step; ++s) {
		float s_up = 0, s_down = sigmas[s+1];
		if (s == 0) {                                                         /* sampling.c:129-137 */
			if (S->have_lmask && mlsd_memcpy(S->d_x0, S->d_x, (size_t)nel*4, 2, st)) return -1;
			const float *nz = noise_draw(S, S->k_draw++);
			if (!nz || mlsd_noise_add_s(S->d_x, nz, sigmas[0], nel, st)) return -1;
			if (S->have_lmask && mlsd_mask_apply(S->d_x, S->d_x0, S->d_lmask, S->hw, nel, st)) return -1;
		}
		if (S->c.s_noise > 0 && s > 0) {                                      /* :139-151 */
			float s_curr = sigmas[s], s_hat = s_curr * sqrt(2) * S->c.s_noise, s_n = sqrt(s_hat*s_hat - s_curr*s_curr);
			const float *nz = noise_draw(S, S->k_draw++);
			if (!nz || mlsd_noise_add_s(S->d_x, nz, s_n, nel, st)) return -1;
			if (S->have_lmask && mlsd_mask_apply(S->d_x, S->d_x0, S->d_lmask, S->hw, nel, st)) return -1;
			solver_t = s_hat;
		}
		if (S->c.s_ancestral > 0) dnsamp_ancestral(sigmas[s], sigmas[s+1], S->c.s_ancestral, &s_down, &s_up);   /* :153-166 */
		const int anc_noise = (s_up > 0 && s+1 != n_step);
		/* draws to have ready by the end of this step's first evaluation: the ancestral one and the next step's s_noise one */
		int k_prefetch = S->k_draw - 1 + anc_noise + ((S->c.s_noise > 0 && s+1 < n_step) ? 1 : 0);
		/* ... and ONE STEP FURTHER: the draws of step s+1 are generated and submitted while evaluation s is being enqueued.  A draw submitted after an evaluation's
		 * uploads (weight streaming) sits behind them in the copy engine's queue -- behind the next evaluation's first segments, which wait for the end of this one:
		 * the noise arrived one upload (10 ms) after the evaluation that needed it (MLSD_ENGINE_TRACE: gaps in front of 4 of 20 evaluations).  Exactly the draws the
		 * schedule will consume: the generator state at the end of the call does not change. */
		if (s + 1 < n_step) {
			float sd2 = sigmas[s+2], su2 = 0;
			if (S->c.s_ancestral > 0) dnsamp_ancestral(sigmas[s+1], sigmas[s+2], S->c.s_ancestral, &sd2, &su2);
			k_prefetch += ((su2 > 0 && s+2 != n_step) ? 1 : 0) + ((S->c.s_noise > 0 && s+2 < n_step) ? 1 : 0);
		}

		/* ---- solver_step(&S->solver, s_down, x)  (src/solvers.c:43-52) */
		const float t0 = solver_t, t1 = s_down;
		if (!(t0 >= 0)) return fail("negative solver time");
		switch (method) {
		case SOLVER_METHOD_EULER: {                                           /* :82-88 */
			if (unet_eval(S, S->d_x, t0, k_prefetch) < 0) return -1;
			if (!P->vparam) {   /* fused: CFG mix, x += dx*dt and the ancestral noise of sampling.c:170-172 in one launch */
				int64_t ld = 0;
				const float *eps = mlctx_tensor_device_f32(S->unet_ctx, S->unet.t_out, &ld);
				const float *nz = anc_noise ? noise_draw(S, S->k_draw) : NULL;
				if (anc_noise && !nz) return -1;
				if (mlsd_euler_cfg_update(S->d_x, eps, ld, B, 4, S->hw, S->c.cfg_scale, t1 - t0, nz, s_up, st)) return -1;
				if (nz) { S->k_draw++; s_up = -1; }                               /* noise already added */
			} else {
				if (dxdt_finish(S, S->d_x, t0, S->d_dx) < 0) return -1;
				if (mlsd_vec_axpy(S->d_x, S->d_x, S->d_dx, t1 - t0, nel, st)) return -1;
			}
		} break;
		case SOLVER_METHOD_HEUN: {                                            /* :96-117 */
			const float dt = t1 - t0;
			float *x1 = S->d_tmp[0], *d1 = S->d_tmp[1];
			if (unet_eval(S, S->d_x, t0, k_prefetch) < 0 || dxdt_finish(S, S->d_x, t0, S->d_dx) < 0) return -1;
			if (mlsd_vec_axpy(x1, S->d_x, S->d_dx, dt, nel, st)) return -1;
			if (!(t1 > 0)) { if (mlsd_memcpy(S->d_x, x1, (size_t)nel*4, 2, st)) return -1; }
			else {
				if (unet_eval(S, x1, t1, -1) < 0 || dxdt_finish(S, x1, t1, d1) < 0) return -1;
				if (mlsd_solver_heun_corr(S->d_x, S->d_dx, d1, dt, nel, st)) return -1;
			}
		} break;
		case SOLVER_METHOD_TAYLOR3: {                                         /* :137-168 */
			const float dt = t1 - t0;
			if (unet_eval(S, S->d_x, t0, k_prefetch) < 0 || dxdt_finish(S, S->d_x, t0, S->d_dx) < 0) return -1;
			float idtp = i_step_solver >= 1 ? 1 / dt_prev : 0, f2 = i_step_solver >= 1 ? dt*dt/2 : 0,
			      f3 = i_step_solver >= 2 ? dt*dt*dt/6 : 0;
			if (mlsd_solver_taylor3(S->d_x, S->d_dx, S->d_tmp[0], S->d_tmp[1], dt, idtp, f2, f3, nel, st)) return -1;
			dt_prev = dt;
		} break;
		case SOLVER_METHOD_DPMPP2M: {                                         /* :207-233 */
			float a = t1 / t0, h = -log(a), c = h / (2*h_last);
			if (i_step_solver == 0 || !(t1 > 0)) c = 0;
			if (unet_eval(S, S->d_x, t0, k_prefetch) < 0 || dxdt_finish(S, S->d_x, t0, S->d_dx) < 0) return -1;
			if (mlsd_solver_dpmpp2m(S->d_x, S->d_dx, S->d_tmp[0], t0, a, c, nel, st)) return -1;
			h_last = h;
		} break;
		case SOLVER_METHOD_DPMPP2S: {                                         /* :264-289 */
			float *x1 = S->d_tmp[0], *dx1 = S->d_tmp[1];
			if (unet_eval(S, S->d_x, t0, k_prefetch) < 0 || dxdt_finish(S, S->d_x, t0, S->d_dx) < 0) return -1;
			if (!(t1 > 0)) { if (mlsd_vec_axpy(S->d_x, S->d_x, S->d_dx, t1 - t0, nel, st)) return -1; }
			else {
				float tm = sqrt(t1 * t0), dt1 = tm - t0, a = t1 / t0;
				if (mlsd_vec_axpy(x1, S->d_x, S->d_dx, dt1, nel, st)) return -1;
				if (unet_eval(S, x1, tm, -1) < 0 || dxdt_finish(S, x1, tm, dx1) < 0) return -1;
				if (mlsd_solver_dpmpp2s(S->d_x, x1, dx1, tm, a, nel, st)) return -1;
			}
		} break;
		default: return fail("invalid sampling method");
		}
		solver_t = t1; i_step_solver++;                                       /* solvers.c:49-50 */

		if (s_up > 0 && s+1 != n_step) {                                      /* sampling.c:170-174 */
			const float *nz = noise_draw(S, S->k_draw++);
			if (!nz || mlsd_noise_add_s(S->d_x, nz, s_up, nel, st)) return -1;
		}
		if (anc_noise) solver_t = sigmas[s+1];
		if (S->have_lmask && mlsd_mask_apply(S->d_x, S->d_x0, S->d_lmask, S->hw, nel, st)) return -1;   /* :176-178 */

		if (S->cb) {   /* mlis_callback (mlimgsynth.c:1741-1745): reports COMPLETED steps, a negative return aborts */
			if (mlsd_stream_sync(st)) return -1;
			int r = S->cb(S->cb_user, s + 1, n_step, S->last_nfe);
			if (r < 0) return r;
		}
	}
	int32_t nan_count = 0;
	if (mlsd_memcpy(&nan_count, S->d_nan, 4, 1, st) || mlsd_stream_sync(st)) return -1;
	if (mlctx_handoff_check(S->unet_ctx) < 0) {                              /* an in-launch hand-off of the plan gave up waiting: results invalid (mlis_amd_denoise retries) */
		if (retry_wanted) *retry_wanted = 1;                                 /* out of band: a user callback may abort with any negative value, -8 included */
		return -1;
	}
	float tot = 0;
	for (int i=0; i<S->i_eval; ++i) { float ms = 0; mlsd_event_elapsed_ms(S->ev[i][0], S->ev[i][1], &ms); tot += ms; }
	{	/* diagnostics (MLSD_ENGINE_TRACE=1): per evaluation, its duration on the device and the gap to the next one */
		static int tr = -1;
		if (tr < 0) { const char *e = getenv("MLSD_ENGINE_TRACE"); tr = e && *e && *e != '0'; }
		if (tr) for (int i=0; i<S->i_eval; ++i) {
			float d = 0, g = 0, t0 = 0;
			mlsd_event_elapsed_ms(S->ev[0][0], S->ev[i][0], &t0); mlsd_event_elapsed_ms(S->ev[i][0], S->ev[i][1], &d);
			if (i + 1 < S->i_eval) mlsd_event_elapsed_ms(S->ev[i][1], S->ev[i+1][0], &g);
			fprintf(stderr, "[engine] evaluation %2d: starts at %8.2f ms, runs %7.2f ms, gap to the next %6.2f ms\n", i, t0, d, g);
		}
	}
	S->last_unet_ms = tot;
	S->unet.nfe += S->last_nfe;
	S->have_init_latent = 0;                /* f_t_ini / latent use flags are cleared after a generation (mlimgsynth.c:700-706) */
	if (nan_count) return mlsd_set_error(-7 /* MLIS_E_NAN */, "NaN found in UNet output (%d values)", nan_count);
	return 1;
}

/* denoise_body + the clean-up of a call that ends early (callback abort, error): the one-step-ahead prefetch has generated -- and advanced the Philox states past -- draws
 * the loop never consumed, and their uploads may still be pending while the next call rewrites h_noise.  The upload stream is drained and every image's
 * generator is rewound to the draws actually consumed, so a following call with seeds = NULL continues the stream where this one stopped. */
static int denoise_once(MLIS_AmdCtx* S, const uint64_t* seeds, int* retry_wanted)
{
	const int r = denoise_body(S, seeds, retry_wanted);
	if (r < 0) {
		if (S->up_stream) mlsd_stream_sync(S->up_stream);
		const int ahead = S->n_draw_gen - S->k_draw;
		if (ahead > 0) for (int b=0;b<S->B;++b) S->rng[b].offset -= (uint32_t)ahead;
		S->n_draw_gen = S->k_draw;
	}
	return r;
}

/* The denoising loop with ONE retry on the hand-off-free plan: a timed-out in-launch hand-off (stream-K slab, LayerNorm statistics: a partner block
 * was not resident -- CUs shared with another process, CU-masked stream) is detected when the results are read back; the loop is deterministic in (initial latent,
 * Philox states, conditioning), so those are kept and the whole loop runs again with plain tiles and separate LayerNorm launches.  Never by restarting the process. */
MLB_API int mlis_amd_denoise(MLIS_AmdCtx* S, const uint64_t* seeds)
{
	if (seeds) mlis_amd_seed(S, seeds);
	const int guard = mlctx_handoff_ops(S->unet_ctx) > 0;
	const int had_init = S->have_init_latent;
	const size_t xbytes = (size_t)S->B * 4 * S->hw * 4;
	RngPhilox keep[MAX_BATCH];
	if (guard) {
		memcpy(keep, S->rng, sizeof(S->rng[0]) * (size_t)S->B);
		if (had_init) {
			if (!S->d_xsave && mlsd_malloc((void**)&S->d_xsave, xbytes)) return -1;
			if (mlsd_memcpy(S->d_xsave, S->d_x, xbytes, 2, S->stream)) return -1;
		}
	}
	int retry = 0;
	int r = denoise_once(S, NULL, &retry);
	if (!retry || !guard) return r;
	mlctx_handoffs_off(S->unet_ctx);
	S->n_handoff_retries++;
	memcpy(S->rng, keep, sizeof(S->rng[0]) * (size_t)S->B);
	S->have_init_latent = had_init;
	if (had_init && mlsd_memcpy(S->d_x, S->d_xsave, xbytes, 2, S->stream)) return -1;
	retry = 0;
	return denoise_once(S, NULL, &retry);
}

/* ------------------------------------------------------------------ VAE tiling
 * sdvae_decode / sdvae_encode with tile_px > 0 (src/vae.c:333-391, 245-300): the latent (image) is cut into overlapping
 * tiles of tile_px (+ a margin of k = 8 latent pixels / 64 image pixels on every side), each tile runs through a tile-sized
 * plan, and the interior of its result is pasted into the output in the reference's order (later tiles overwrite the
 * margins of earlier ones).  TAE codecs are never tiled (the reference only tiles sdvae_*). */
MLB_API int mlis_amd_set_vae_tile(MLIS_AmdCtx* S, int tile_px)
{
	if (tile_px < 0) tile_px = 0;
	if (tile_px != S->vae_tile) {
		mlsd_stream_sync(S->stream);
		if (S->dect_ctx) { mlctx_destroy(S->dect_ctx); S->dect_ctx = NULL; }
		if (S->enct_ctx) { mlctx_destroy(S->enct_ctx); S->enct_ctx = NULL; }
		S->vae_tile = tile_px;
	}
	return 1;
}

static int tile_dims(int tile_px, int margin2, int full, int unit)
{	/* n = min(round_up(tile_px, 64)/unit + 2k, full); returns full when one tile covers everything (tiling disabled) */
	const int t = (tile_px + 63) / 64 * 64;
	const int n = t / unit + margin2;
	return n < full ? n : full;
}

/* builds (once) the decode tile plan; returns NULL with no error when tiling does not apply */
MLB_API MLCtx* mlis_amd_decoder_tile_prepare(MLIS_AmdCtx* S)
{
	if (S->dect_ctx) return S->dect_ctx;
	if (S->vae_tile <= 0 || S->c.use_tae) return NULL;
	const int f = S->vae_p.f_down, k = 8;
	const int n0 = tile_dims(S->vae_tile, 2*k, S->lw, f), n1 = tile_dims(S->vae_tile, 2*k, S->lh, f);
	if (n0 == S->lw && n1 == S->lh) return NULL;                                      /* one tile: disabled (vae.c:347-348) */
	const int B = S->B;
	if (!S->d_lat_tile && mlsd_malloc((void**)&S->d_lat_tile, (size_t)B*4*n0*n1*4)) return NULL;
	if (!S->d_img_tile && mlsd_malloc((void**)&S->d_img_tile, (size_t)B*3*n0*f*n1*f*4)) return NULL;
	MLCtx *C = mlctx_new(S->stream);
	int ok = sdvae_decode_init(C, &S->vae_p, n0, n1, B, &S->t_lat_dect) >= 0 &&
	         mlctx_input_bind(S->t_lat_dect, S->d_lat_tile, B, NULL, 1 / S->vae_p.scale_factor, 0) >= 0 &&
	         sdvae_decode_build(C, &S->vae_p, S->t_lat_dect) >= 0;
	if (ok && !S->c.defer_weights) ok = mlctx_params_synth(C, S->c.weight_seed) >= 0;
	if (!ok) { mlctx_destroy(C); return NULL; }
	S->dect_ctx = C; S->tl_w = n0; S->tl_h = n1;
	return C;
}

MLB_API MLCtx* mlis_amd_encoder_tile_prepare(MLIS_AmdCtx* S)
{
	if (S->enct_ctx) return S->enct_ctx;
	if (S->vae_tile <= 0 || S->c.use_tae) return NULL;
	const int f = S->vae_p.f_down, k = f*8, W = S->c.width, H = S->c.height;
	const int n0 = tile_dims(S->vae_tile, 2*k, W, 1), n1 = tile_dims(S->vae_tile, 2*k, H, 1);
	if (n0 == W && n1 == H) return NULL;
	const int B = S->B;
	if (!S->d_img_in && mlsd_malloc((void**)&S->d_img_in, (size_t)B*3*W*H*4)) return NULL;
	if (!S->d_imgin_tile && mlsd_malloc((void**)&S->d_imgin_tile, (size_t)B*3*n0*n1*4)) return NULL;
	if (!S->d_mom_tile && mlsd_malloc((void**)&S->d_mom_tile, (size_t)B*8*(n0/f)*(n1/f)*4)) return NULL;
	if (!S->d_mom && mlsd_malloc((void**)&S->d_mom, (size_t)B*8*S->hw*4)) return NULL;
	MLCtx *C = mlctx_new(S->stream);
	int ok = sdvae_encode_init(C, &S->vae_p, n0, n1, B, &S->t_img_enct) >= 0 &&
	         mlctx_input_bind(S->t_img_enct, S->d_imgin_tile, B, NULL, 1.0f, 2) >= 0 &&
	         sdvae_encode_build(C, &S->vae_p, S->t_img_enct) >= 0;
	if (ok && !S->c.defer_weights) ok = mlctx_params_synth(C, S->c.weight_seed) >= 0;
	if (!ok) { mlctx_destroy(C); return NULL; }
	S->enct_ctx = C; S->ti_w = n0; S->ti_h = n1;
	return C;
}

static int decode_tiled(MLIS_AmdCtx* S)
{	/* src/vae.c:351-383 */
	const int f = S->vae_p.f_down, k = 8, B = S->B;
	const int lat_n0 = S->lw, lat_n1 = S->lh, n0 = S->tl_w, n1 = S->tl_h;
	const int img_n0 = lat_n0*f, img_n1 = lat_n1*f;
	const int step0 = n0 - 2*k, step1 = n1 - 2*k;
	const int n_tile0 = (lat_n0 + step0 - 1) / step0, n_tile1 = (lat_n1 + step1 - 1) / step1;
	void *st = S->stream;
	MLTensor *r = mlctx_result(S->dect_ctx);
	int64_t ld = 0;
	const float *y = mlctx_tensor_device_f32(S->dect_ctx, r, &ld);
	for (int t1=0; t1<n_tile1; ++t1) {
		const int i1 = t1*step1 < lat_n1 - n1 ? t1*step1 : lat_n1 - n1;
		for (int t0=0; t0<n_tile0; ++t0) {
			const int i0 = t0*step0 < lat_n0 - n0 ? t0*step0 : lat_n0 - n0;
			if (mlsd_copy_slice2(S->d_lat_tile, n0, n1, S->d_x, lat_n0, lat_n1, n0, n1, 0, 0, i0, i1, B*4, st)) return -1;
			if (mlctx_compute(S->dect_ctx) < 0) return -1;
			if (mlsd_nhwc_to_nchw_f32(y, ld, B, 3, n0*f*n1*f, S->d_img_tile, 0.5f, 0.5f, st)) return -1;   /* (x+1)/2: elementwise, commutes with the paste */
			const int d0 = i0 ? k : 0, d1 = i1 ? k : 0;
			if (mlsd_copy_slice2(S->d_img, img_n0, img_n1, S->d_img_tile, n0*f, n1*f, (n0-k)*f, (n1-k)*f, (i0+d0)*f, (i1+d1)*f,
					d0*f, d1*f, B*3, st)) return -1;
		}
	}
	return 1;
}

static int encode_tiled(MLIS_AmdCtx* S)
{	/* src/vae.c:262-299: tiles of the image -> moments tiles -> pasted into the full moments tensor (NCHW [B][8][lh][lw]) */
	const int f = S->vae_p.f_down, k = f*8, B = S->B;
	const int img_n0 = S->c.width, img_n1 = S->c.height, n0 = S->ti_w, n1 = S->ti_h;
	const int lat_n0 = img_n0/f, lat_n1 = img_n1/f;
	const int step0 = n0 - 2*k, step1 = n1 - 2*k;
	const int n_tile0 = (img_n0 + step0 - 1) / step0, n_tile1 = (img_n1 + step1 - 1) / step1;
	void *st = S->stream;
	MLTensor *r = mlctx_result(S->enct_ctx);
	int64_t ld = 0;
	const float *y = mlctx_tensor_device_f32(S->enct_ctx, r, &ld);
	for (int t1=0; t1<n_tile1; ++t1) {
		const int i1 = t1*step1 < img_n1 - n1 ? t1*step1 : img_n1 - n1;
		for (int t0=0; t0<n_tile0; ++t0) {
			const int i0 = t0*step0 < img_n0 - n0 ? t0*step0 : img_n0 - n0;
			if (mlsd_copy_slice2(S->d_imgin_tile, n0, n1, S->d_img_in, img_n0, img_n1, n0, n1, 0, 0, i0, i1, B*3, st)) return -1;
			if (mlctx_compute(S->enct_ctx) < 0) return -1;
			if (mlsd_count_nonfinite(y, (size_t)B*(n0/f)*(n1/f)*ld, S->d_nan, st)) return -1;
			if (mlsd_nhwc_to_nchw_f32(y, ld, B, 8, (n0/f)*(n1/f), S->d_mom_tile, 1.0f, 0.0f, st)) return -1;
			const int d0 = i0 ? k : 0, d1 = i1 ? k : 0;
			if (mlsd_copy_slice2(S->d_mom, lat_n0, lat_n1, S->d_mom_tile, n0/f, n1/f, (n0-k)/f, (n1-k)/f, (i0+d0)/f, (i1+d1)/f,
					d0/f, d1/f, B*8, st)) return -1;
		}
	}
	return 1;
}

/* after the launches of a decode / encode were enqueued: if the plan hands data over inside launches, drain and check; 1 = clean, 0 = gave up -> the plan is now
 * hand-off-free and the caller runs its launches again, < 0 = error */
static int handoffs_clean(MLIS_AmdCtx* S, MLCtx* C, int attempt)
{
	if (!mlctx_handoff_ops(C) && !attempt) return 1;
	if (mlsd_stream_sync(S->stream)) return -1;
	if (mlctx_handoff_check(C) == 0) return 1;
	if (attempt) return -1;
	mlctx_handoffs_off(C);
	S->n_handoff_retries++;
	return 0;
}

MLB_API int mlis_amd_decode(MLIS_AmdCtx* S)
{
	if (S->vae_tile > 0 && !S->c.use_tae && mlis_amd_decoder_tile_prepare(S)) {
		for (int attempt=0; attempt<2; ++attempt) {
			if (decode_tiled(S) < 0) return -1;
			const int ok = handoffs_clean(S, S->dect_ctx, attempt);
			if (ok) return ok;
		}
		return -1;
	}
	for (int attempt=0; ; ++attempt) {
		if (mlctx_compute(S->dec_ctx) < 0) return -1;
		const int ok = handoffs_clean(S, S->dec_ctx, attempt);
		if (ok < 0) return -1;
		if (ok) break;
	}
	int64_t ld = 0;
	MLTensor *r = mlctx_result(S->dec_ctx);
	const float *y = mlctx_tensor_device_f32(S->dec_ctx, r, &ld);
	const int HW = S->c.width * S->c.height;
	/* sdvae_decoder_post (x+1)/2 (vae.h:43-47); TAE output is used as-is */
	const float mul = S->c.use_tae ? 1.0f : 0.5f, add = S->c.use_tae ? 0.0f : 0.5f;
	if (mlsd_nhwc_to_nchw_f32(y, ld, S->B, 3, HW, S->d_img, mul, add, S->stream)) return -1;
	return 1;
}

/* builds the encoder plan (once) without running it; with defer_weights the caller loads its parameters before the first encode */
MLB_API MLCtx* mlis_amd_encoder_prepare(MLIS_AmdCtx* S)
{
	if (S->enc_ctx) return S->enc_ctx;
	const int B = S->B, W = S->c.width, H = S->c.height;
	void *st = S->stream;
	if (!S->d_img_in && mlsd_malloc((void**)&S->d_img_in, (size_t)B*3*W*H*4)) return NULL;
	MLCtx *C = mlctx_new(st);
	int ok = 0;
	if (S->c.use_tae) {
		ok = sdtae_encode_init(C, W, H, B, &S->t_img_enc) >= 0 && mlctx_input_bind(S->t_img_enc, S->d_img_in, B, NULL, 1.0f, 0) >= 0 &&
		     sdtae_encode_build(C, S->t_img_enc) >= 0;
	} else {
		ok = sdvae_encode_init(C, &S->vae_p, W, H, B, &S->t_img_enc) >= 0 &&
		     mlctx_input_bind(S->t_img_enc, S->d_img_in, B, NULL, 1.0f, 2) >= 0 &&       /* mode 2: x*2-1 (vae.h:36-40) */
		     sdvae_encode_build(C, &S->vae_p, S->t_img_enc) >= 0;
	}
	if (ok && !S->c.defer_weights) ok = mlctx_params_synth(C, S->c.weight_seed) >= 0;
	if (!ok) { mlctx_destroy(C); return NULL; }
	S->enc_ctx = C;
	return C;
}

/* mlis_image_encode (src/mlimgsynth.c:1301-1330): images host NCHW [B][3][H][W] in [0,1] -> the resident latent becomes
 * the sampled (VAE: mean + std*N(0,1), one Philox call per image, src/vae.c:203-229) or direct (TAE) encoding and is
 * marked as the initial latent of the next denoise.  The encoder plan is built on first use. */
MLB_API int mlis_amd_encode(MLIS_AmdCtx* S, const float* images, int sample)
{
	const int B = S->B, W = S->c.width, H = S->c.height;
	const size_t img_elems = (size_t)B*3*W*H;
	void *st = S->stream;
	const int tiled = S->vae_tile > 0 && !S->c.use_tae && mlis_amd_encoder_tile_prepare(S) != NULL;
	if (!tiled && !S->enc_ctx && !mlis_amd_encoder_prepare(S)) return -1;
	if (mlsd_memcpy(S->d_img_in, images, img_elems*4, 0, st)) return -1;
	if (mlsd_memset(S->d_nan, 0, 4, st)) return -1;
	const float *y; int64_t ld = 0;
	for (int attempt=0; ; ++attempt) {            /* (second pass only after a timed-out in-launch hand-off: the plan is hand-off-free then) */
		if (attempt && mlsd_memset(S->d_nan, 0, 4, st)) return -1;
		if (tiled) {
			if (encode_tiled(S) < 0) return -1;
			y = NULL;
		} else {
			if (mlctx_compute(S->enc_ctx) < 0) return -1;
			y = mlctx_tensor_device_f32(S->enc_ctx, mlctx_result(S->enc_ctx), &ld);
			if (mlsd_count_nonfinite(y, (size_t)B*S->hw*ld, S->d_nan, st)) return -1;
		}
		const int ok = handoffs_clean(S, tiled ? S->enct_ctx : S->enc_ctx, attempt);
		if (ok < 0) return -1;
		if (ok) break;
	}
	if (S->c.use_tae) {
		if (mlsd_nhwc_to_nchw_f32(y, ld, B, 4, S->hw, S->d_x, 1.0f, 0.0f, st)) return -1;
	} else {
		const float *rnd = NULL;
		if (sample) {   /* sdvae_latent_sample: rng_randn(n) per image, before any denoising draw */
			S->n_draw_gen = 0;
			rnd = noise_draw(S, 0);
			if (!rnd) return -1;
		}
		if (tiled) { if (mlsd_latent_sample_nchw(S->d_mom, rnd, S->d_x, B, S->vae_p.ch_z, S->hw, S->vae_p.scale_factor, st)) return -1; }
		else if (mlsd_latent_sample(y, ld, rnd, S->d_x, B, S->vae_p.ch_z, S->hw, S->vae_p.scale_factor, st)) return -1;
	}
	int32_t nan_count = 0;
	if (mlsd_memcpy(&nan_count, S->d_nan, 4, 1, st) || mlsd_stream_sync(st)) return -1;
	if (nan_count) return mlsd_set_error(-7, "NaN found in encoded latent");   /* mlimgsynth.c:1323 */
	S->have_init_latent = 1;
	return 1;
}

MLB_API int mlis_amd_generate(MLIS_AmdCtx* S, const uint64_t* seeds, float* latents_out, float* images_out)
{
	int r = mlis_amd_denoise(S, seeds);
	if (r < 0) return r;
	if (latents_out && mlsd_memcpy(latents_out, S->d_x, (size_t)S->B*4*S->hw*4, 1, S->stream)) return -1;
	if (mlis_amd_decode(S) < 0) return -1;
	if (images_out && mlsd_memcpy(images_out, S->d_img, (size_t)S->B*3*S->c.width*S->c.height*4, 1, S->stream)) return -1;
	if (mlsd_stream_sync(S->stream)) return -1;
	return 1;
}

MLB_API int mlis_amd_handoff_retries(const MLIS_AmdCtx* S) { return S->n_handoff_retries; }   /* denoise / decode / encode passes re-run on the hand-off-free plan */

MLB_API int mlis_amd_sync(MLIS_AmdCtx* S) { return mlsd_stream_sync(S->stream) ? -1 : 1; }
MLB_API void* mlis_amd_latent_device(MLIS_AmdCtx* S) { return S->d_x; }
MLB_API void* mlis_amd_image_device(MLIS_AmdCtx* S) { return S->d_img; }
MLB_API MLCtx* mlis_amd_unet_ctx(MLIS_AmdCtx* S) { return S->unet_ctx; }
MLB_API MLCtx* mlis_amd_decoder_ctx(MLIS_AmdCtx* S) { return S->dec_ctx; }
MLB_API MLCtx* mlis_amd_encoder_ctx(MLIS_AmdCtx* S) { return S->enc_ctx; }
MLB_API float mlis_amd_last_unet_ms(MLIS_AmdCtx* S) { return S->last_unet_ms; }
MLB_API int mlis_amd_last_nfe(MLIS_AmdCtx* S) { return S->last_nfe; }
MLB_API int mlis_amd_last_n_step(MLIS_AmdCtx* S) { return S->last_n_step; }

MLB_API int mlis_amd_info(MLIS_AmdCtx* S, double* unet_flops, double* dec_flops, int* unet_ops, size_t* mem_params, size_t* mem_compute)
{
	MLCtxInfo a, b;
	mlctx_info(S->unet_ctx, &a); mlctx_info(S->dec_ctx, &b);
	if (unet_flops) *unet_flops = a.flops;
	if (dec_flops) *dec_flops = b.flops;
	if (unet_ops) *unet_ops = (int)a.n_ops;
	if (mem_params) *mem_params = a.mem_params + b.mem_params;
	if (mem_compute) *mem_compute = a.mem_compute + b.mem_compute;
	return 1;
}
