/* Generation driver — the slice of the reference's mlis_generate (src/mlimgsynth.c:1634-1773) that
 * drives the hot path: schedule, initial noise, 20 x [CFG UNet evaluation + Euler(-ancestral) update],
 * latent decode.  MI355X-first differences:
 *   - n_batch images per GPU; cond and uncond of all images form ONE batch-2B UNet evaluation
 *     (the reference runs two sequential batch-1 evaluations per step, mlimgsynth.c:1578-1582);
 *   - the latent never leaves the device: c_in scaling happens in the UNet's input conversion,
 *     the CFG mix + Euler step + ancestral noise are one elementwise kernel;
 *   - all per-step scalars (t, c_in, dt, sigma_up) are computed up front on the host with the
 *     reference's own formulas and uploaded once; the Philox/Box-Muller noise is generated on the host
 *     (bit-exact integer stream, fp64 Box-Muller like src/ccommon/rng_philox.c) WHILE the GPU runs the
 *     UNet evaluation of the same step, then copied asynchronously;
 *   - weights stay resident in HBM (288 GB) for the lifetime of the context instead of being
 *     re-uploaded per generate (mlblock.c:266-292) or per half-graph (--unet-split, unet.c:390-458).
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>

int unet_denoise_build(UnetState* S);

#define MAX_STEPS 256

struct MLIS_AmdCtx {
	MLIS_AmdConfig c;
	char model[16];
	void* stream;
	UnetParams unet_p;
	VaeParams vae_p;
	int lw, lh, hw, B, N;
	MLCtx *unet_ctx, *dec_ctx;
	UnetState unet;
	MLTensor *t_lat_dec;
	/* device state */
	float *d_x;                 /* latent [B][4][hw] NCHW fp32 */
	float *d_img;               /* image  [B][3][H][W] fp32 */
	float *d_cin;               /* [B] current c_in (read by the UNet input conversion) */
	float *d_tall, *d_cinall, *d_dtall, *d_supall, *d_sig0;   /* per-step scalar tables */
	float *d_noise;             /* [n_step+1][B][4*hw] */
	float *h_noise;             /* pinned, same shape */
	float *h_scal;              /* pinned staging of the scalar tables */
	int32_t *d_nan;
	void *ev[MAX_STEPS][2];
	int n_ev;
	float last_unet_ms;
	int last_nfe;
	int cond_set;
	int own_stream;
};

static int fail(const char* msg) { return mlsd_set_error(-1, "%s", msg); }

MLB_API void mlis_amd_destroy(MLIS_AmdCtx* S)
{
	if (!S) return;
	mlsd_stream_sync(S->stream);
	if (S->unet_ctx) mlctx_destroy(S->unet_ctx);
	if (S->dec_ctx) mlctx_destroy(S->dec_ctx);
	mlsd_free(S->d_x); mlsd_free(S->d_img); mlsd_free(S->d_cin); mlsd_free(S->d_tall); mlsd_free(S->d_cinall);
	mlsd_free(S->d_dtall); mlsd_free(S->d_supall); mlsd_free(S->d_sig0); mlsd_free(S->d_noise); mlsd_free(S->d_nan);
	mlsd_host_free(S->h_noise); mlsd_host_free(S->h_scal);
	for (int i=0;i<S->n_ev;++i) { mlsd_event_destroy(S->ev[i][0]); mlsd_event_destroy(S->ev[i][1]); }
	if (S->own_stream) mlsd_stream_destroy(S->stream);
	free(S);
}

MLB_API MLIS_AmdCtx* mlis_amd_create(const MLIS_AmdConfig* cfg, void* stream)
{
	MLIS_AmdCtx *S = (MLIS_AmdCtx*)calloc(1, sizeof(*S));
	S->c = *cfg; S->stream = stream;
	if (!stream) {   /* own non-default stream: stream capture (hipGraph) is not permitted on the NULL stream */
		if (mlsd_stream_create(&S->stream)) { free(S); return NULL; }
		S->own_stream = 1;
		stream = S->stream;
	}
	snprintf(S->model, sizeof(S->model), "%s", cfg->model ? cfg->model : "sd1");
	S->c.model = S->model;
	if (S->c.n_step < 1) S->c.n_step = 20;
	if (S->c.n_step >= MAX_STEPS) { fail("too many steps"); goto err; }
	if (S->c.n_batch < 1) S->c.n_batch = 1;
	if (!(S->c.cfg_scale > 0)) S->c.cfg_scale = 7;          /* default cfg 7, src/mlimgsynth.c:474 */
	if (!S->c.sched) S->c.sched = DNSAMP_SCHED_UNIFORM;
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
	if (mlsd_malloc((void**)&S->d_img, (size_t)B*3*S->c.width*S->c.height*4)) goto err;
	if (mlsd_malloc((void**)&S->d_cin, B*4)) goto err;
	if (mlsd_malloc((void**)&S->d_tall, (size_t)ns*N*4)) goto err;
	if (mlsd_malloc((void**)&S->d_cinall, (size_t)ns*B*4)) goto err;
	if (mlsd_malloc((void**)&S->d_dtall, (size_t)ns*B*4)) goto err;
	if (mlsd_malloc((void**)&S->d_supall, (size_t)ns*B*4)) goto err;
	if (mlsd_malloc((void**)&S->d_sig0, B*4)) goto err;
	if (mlsd_malloc((void**)&S->d_noise, (size_t)(ns+1)*lat_elems*4)) goto err;
	if (mlsd_malloc((void**)&S->d_nan, 256)) goto err;
	if (mlsd_host_alloc((void**)&S->h_noise, (size_t)(ns+1)*lat_elems*4)) goto err;
	if (mlsd_host_alloc((void**)&S->h_scal, (size_t)ns*(N+3*B)*4 + B*4)) goto err;
	for (int i=0;i<ns;++i) { if (mlsd_event_create(&S->ev[i][0]) || mlsd_event_create(&S->ev[i][1])) goto err; S->n_ev = i+1; }

	/* ---- UNet plan, x bound to the resident latent (c_in scaling + cond/uncond duplication in the gather) */
	S->unet_ctx = mlctx_new(stream);
	if (S->c.use_hipgraph) mlctx_set_flags(S->unet_ctx, MLB_F_HIPGRAPH);
	if (unet_denoise_init(&S->unet, S->unet_ctx, &S->unet_p, S->lw, S->lh, N) < 0) goto err;
	if (mlctx_input_bind(S->unet.t_x, S->d_x, B, S->d_cin, 1.0f, 0) < 0) { fail("input bind failed"); goto err; }
	if (unet_denoise_build(&S->unet) < 0) goto err;
	if (mlctx_params_synth(S->unet_ctx, S->c.weight_seed) < 0) goto err;

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
	if (mlctx_params_synth(S->dec_ctx, S->c.weight_seed) < 0) goto err;
	return S;
err:
	mlis_amd_destroy(S);
	return NULL;
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

MLB_API int mlis_amd_denoise(MLIS_AmdCtx* S, const uint64_t* seeds)
{
	if (!S->cond_set) return fail("mlis_amd_denoise: conditioning not set");
	const UnetParams *P = &S->unet_p;
	const int B = S->B, N = S->N;
	const size_t per = (size_t)4 * S->hw, lat_elems = (size_t)B * per;
	void *st = S->stream;

	/* ---- dnsamp_init: schedule + per-step scalars, exactly the reference's host arithmetic */
	float sigmas[MAX_STEPS+1];
	const int n_step = dnsamp_schedule(P, S->c.n_step, S->c.sched, 1.0f, 0.0f, sigmas);
	if (n_step < 0 || n_step > S->c.n_step) return fail("schedule failed");
	float *h_t = S->h_scal, *h_cin = h_t + (size_t)n_step*N, *h_dt = h_cin + (size_t)n_step*B,
	      *h_sup = h_dt + (size_t)n_step*B, *h_sig0 = h_sup + (size_t)n_step*B;
	int need_noise[MAX_STEPS];
	float solver_t = sigmas[0];                                   /* sampling.c:92 */
	for (int s=0; s<n_step; ++s) {
		float s_up = 0, s_down = sigmas[s+1];
		if (S->c.s_ancestral > 0) dnsamp_ancestral(sigmas[s], sigmas[s+1], S->c.s_ancestral, &s_down, &s_up);
		const float sigma = solver_t;                              /* dxdt evaluated at the solver's t (solvers.c:84-85) */
		const float t = unet_sigma_to_t(P, sigma);
		const float c_in = 1 / sqrt(sigma*sigma + 1);              /* unet.c:471 */
		const float dt = s_down - sigma;                           /* solvers.c:84 */
		for (int n=0;n<N;++n) h_t[(size_t)s*N + n] = t;
		for (int b=0;b<B;++b) { h_cin[(size_t)s*B+b] = c_in; h_dt[(size_t)s*B+b] = dt; h_sup[(size_t)s*B+b] = s_up; }
		solver_t = s_down;
		need_noise[s] = (s_up > 0 && s+1 != n_step);
		if (need_noise[s]) solver_t = sigmas[s+1];                 /* sampling.c:173 */
	}
	for (int b=0;b<B;++b) h_sig0[b] = sigmas[0];
	if (mlsd_memcpy(S->d_tall, h_t, (size_t)n_step*N*4, 0, st) || mlsd_memcpy(S->d_cinall, h_cin, (size_t)n_step*B*4, 0, st) ||
	    mlsd_memcpy(S->d_dtall, h_dt, (size_t)n_step*B*4, 0, st) || mlsd_memcpy(S->d_supall, h_sup, (size_t)n_step*B*4, 0, st) ||
	    mlsd_memcpy(S->d_sig0, h_sig0, B*4, 0, st)) return -1;

	/* ---- initial latent: zeros + N(0,1)*sigma_0 (mlimgsynth.c:1669-1670, sampling.c:133) */
	RngPhilox rng[64];
	if (B > 64) return fail("n_batch > 64 not supported");
	for (int b=0;b<B;++b) { rng[b].seed = seeds[b]; rng[b].offset = 0; }
	for (int b=0;b<B;++b) rng_philox_randn(&rng[b], (unsigned)per, S->h_noise + (size_t)b*per);
	if (mlsd_memset(S->d_x, 0, lat_elems*4, st) || mlsd_memset(S->d_nan, 0, 4, st)) return -1;
	if (mlsd_memcpy(S->d_noise, S->h_noise, lat_elems*4, 0, st)) return -1;
	if (mlsd_noise_add(S->d_x, S->d_noise, S->d_sig0, B, (int64_t)per, st)) return -1;

	int64_t ld_eps = 0;
	const float *eps = mlctx_tensor_device_f32(S->unet_ctx, S->unet.t_out, &ld_eps);
	float *d_t_in = (float*)mlctx_input_device_ptr(S->unet.t_t);
	int nfe = 0;
	for (int s=0; s<n_step; ++s) {
		/* this step's scalars into the plan's fixed input slots */
		if (mlsd_memcpy(d_t_in, S->d_tall + (size_t)s*N, (size_t)N*4, 2, st) ||
		    mlsd_memcpy(S->d_cin, S->d_cinall + (size_t)s*B, (size_t)B*4, 2, st)) return -1;
		mlsd_event_record(S->ev[s][0], st);
		if (mlctx_compute(S->unet_ctx) < 0) return -1;           /* cond + uncond of all images: one evaluation */
		mlsd_event_record(S->ev[s][1], st);
		nfe += N / B;
		if (mlsd_count_nonfinite(eps, (size_t)N*S->hw*4, S->d_nan, st)) return -1;   /* ltensor_finite_check, unet.c:487 */
		const float *noise = NULL;
		if (need_noise[s]) {
			/* host Philox for THIS step while the GPU is busy with the evaluation just enqueued */
			float *hn = S->h_noise + (size_t)(s+1)*lat_elems, *dn = S->d_noise + (size_t)(s+1)*lat_elems;
			for (int b=0;b<B;++b) rng_philox_randn(&rng[b], (unsigned)per, hn + (size_t)b*per);
			if (mlsd_memcpy(dn, hn, lat_elems*4, 0, st)) return -1;
			noise = dn;
		}
		if (mlsd_sampler_update(S->d_x, eps, ld_eps, B, 4, S->hw, S->c.cfg_scale, S->d_dtall + (size_t)s*B, noise,
				S->d_supall + (size_t)s*B, st)) return -1;
	}
	int32_t nan_count = 0;
	if (mlsd_memcpy(&nan_count, S->d_nan, 4, 1, st) || mlsd_stream_sync(st)) return -1;
	float tot = 0;
	for (int s=0; s<n_step; ++s) { float ms = 0; mlsd_event_elapsed_ms(S->ev[s][0], S->ev[s][1], &ms); tot += ms; }
	S->last_unet_ms = tot; S->last_nfe = nfe;
	S->unet.nfe += nfe;
	if (nan_count) return mlsd_set_error(-1, "NaN found in UNet output (%d values)", nan_count);
	return 1;
}

MLB_API int mlis_amd_decode(MLIS_AmdCtx* S)
{
	if (mlctx_compute(S->dec_ctx) < 0) return -1;
	int64_t ld = 0;
	MLTensor *r = mlctx_result(S->dec_ctx);
	const float *y = mlctx_tensor_device_f32(S->dec_ctx, r, &ld);
	const int HW = S->c.width * S->c.height;
	/* sdvae_decoder_post (x+1)/2 (vae.h:43-47); TAE output is used as-is */
	const float mul = S->c.use_tae ? 1.0f : 0.5f, add = S->c.use_tae ? 0.0f : 0.5f;
	if (mlsd_nhwc_to_nchw_f32(y, ld, S->B, 3, HW, S->d_img, mul, add, S->stream)) return -1;
	return 1;
}

MLB_API int mlis_amd_generate(MLIS_AmdCtx* S, const uint64_t* seeds, float* latents_out, float* images_out)
{
	if (mlis_amd_denoise(S, seeds) < 0) return -1;
	if (latents_out && mlsd_memcpy(latents_out, S->d_x, (size_t)S->B*4*S->hw*4, 1, S->stream)) return -1;
	if (mlis_amd_decode(S) < 0) return -1;
	if (images_out && mlsd_memcpy(images_out, S->d_img, (size_t)S->B*3*S->c.width*S->c.height*4, 1, S->stream)) return -1;
	if (mlsd_stream_sync(S->stream)) return -1;
	return 1;
}

MLB_API void* mlis_amd_latent_device(MLIS_AmdCtx* S) { return S->d_x; }
MLB_API void* mlis_amd_image_device(MLIS_AmdCtx* S) { return S->d_img; }
MLB_API MLCtx* mlis_amd_unet_ctx(MLIS_AmdCtx* S) { return S->unet_ctx; }
MLB_API MLCtx* mlis_amd_decoder_ctx(MLIS_AmdCtx* S) { return S->dec_ctx; }
MLB_API float mlis_amd_last_unet_ms(MLIS_AmdCtx* S) { return S->last_unet_ms; }
MLB_API int mlis_amd_last_nfe(MLIS_AmdCtx* S) { return S->last_nfe; }

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
