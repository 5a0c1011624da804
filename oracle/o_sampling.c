/* ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Host loop restated from the reference:
 *   sigma tables          src/unet.c:283-334
 *   unet_denoise_run      src/unet.c:460-498
 *   scheduler / sampler   src/sampling.c:28-185
 *   Euler solver          src/solvers.c:43-52,82-88
 *   CFG dxdt              src/mlimgsynth.c:1565-1587
 *   Philox randn          src/ccommon/rng_philox.c:9-51   (pinned by src/test_rng.c:11-24)
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <time.h>

/* ------------------------------------------------------------------ sigma tables */
static float g_log_sigmas[1000];
static int g_log_sigmas_init = 0;

static void log_sigmas_init(void)
{	/* unet_params_init, src/unet.c:283-303: scaled-linear betas, all in double */
	if (g_log_sigmas_init) return;
	unsigned n = 1000;
	double linear_start = 0.00085, linear_end = 0.0120,
	       b = sqrt(linear_start), e = sqrt(linear_end),
	       f = (e - b) / (n - 1), alpha_cumprod = 1.0;
	for (unsigned i=0; i<n; ++i) {
		double beta = b + f*i, alpha = 1.0 - beta*beta;
		alpha_cumprod *= alpha;
		double sigma = sqrt((1 - alpha_cumprod) / alpha_cumprod);
		g_log_sigmas[i] = log(sigma);
	}
	g_log_sigmas_init = 1;
}

void orc_log_sigmas(float* out) { log_sigmas_init(); memcpy(out, g_log_sigmas, sizeof(g_log_sigmas)); }

static float linear_interp(unsigned n, const float* vec, float t)
{	/* src/unet.c:305-312 */
	int ti = t;
	if (ti < 0) ti = 0; else if (ti > (int)n-1) ti = (int)n-1;
	float v1 = vec[ti], v2 = ti+1 < (int)n ? vec[ti+1] : v1;
	return v1*(ti+1-t) + v2*(t-ti);
}

static float linear_est(unsigned n, const float* vec, float v)
{	/* src/unet.c:315-322 with BISECT_RIGHT_DECL (src/ccommon/bisect.h:38-51):
	 * the comparison copysign(1, vec[i]-v) is never 0, so idx = first i with vec[i] >= v */
	size_t b=0, e=n;
	while (b < e) {
		size_t i = (b+e)/2;
		int r = (int)copysign(1, vec[i] - v);
		if (r < 0) b = i+1; else e = i;
	}
	size_t idx = b;
	if (idx+1 >= n) return n-1;
	float v1 = vec[idx], v2 = vec[idx+1];
	return idx + (v - v1) / (v2 - v1);
}

float orc_sigma_to_t(float sigma)
{	/* unet_sigma_to_t :324-328 */
	log_sigmas_init();
	float ls = log(sigma);
	return linear_est(1000, g_log_sigmas, ls);
}

float orc_t_to_sigma(float t)
{	/* unet_t_to_sigma :330-334 */
	log_sigmas_init();
	float ls = linear_interp(1000, g_log_sigmas, t);
	return exp(ls);
}

int orc_schedule(int n_step_req, int sched, float f_t_ini, float f_t_end, float* sigmas)
{	/* dnsamp_init, src/sampling.c:28-96 for a 1-NFE solver (Euler) */
	int n_step = n_step_req;
	if (n_step < 1) n_step = 20;
	if (!(f_t_ini > 0)) f_t_ini = 1;
	n_step = n_step * (f_t_ini - f_t_end) + 0.5;
	if (n_step < 1) n_step = 1;
	sigmas[n_step] = 0;
	float t_ini = (1000 - 1) * f_t_ini, t_end = (1000 - 1) * f_t_end;
	if (!sched) sched = 1;
	if (sched == 1) {          /* DNSAMP_SCHED_UNIFORM */
		float b = t_ini, f = n_step>1 ? (t_end-t_ini)/(n_step-1) : 0;
		for (unsigned i=0; i<(unsigned)n_step; ++i) sigmas[i] = orc_t_to_sigma(b+i*f);
	} else {                   /* DNSAMP_SCHED_KARRAS */
		float smin = orc_t_to_sigma(t_end), smax = orc_t_to_sigma(t_ini), p=7,
		      sminp = pow(smin, 1/p), smaxp = pow(smax, 1/p), b = smaxp,
		      f = n_step>1 ? (sminp - smaxp) / (n_step-1) : 0;
		for (unsigned i=0; i<(unsigned)n_step; ++i) sigmas[i] = pow(b+i*f, p);
	}
	return n_step;
}

void orc_ancestral(float s1, float s2, float eta, float* s_down, float* s_up)
{	/* src/sampling.c:153-166 */
	float up = sqrt((s2*s2) * (s1*s1 - s2*s2) / (s1*s1));
	up *= eta;
	if (up > s2) up = s2;
	*s_up = up;
	*s_down = sqrt(s2*s2 - up*up);
}

/* ------------------------------------------------------------------ Philox4x32-10 + Box-Muller */
static const uint32_t philox_m[2] = {0xD2511F53, 0xCD9E8D57};
static const uint32_t philox_w[2] = {0x9E3779B9, 0xBB67AE85};

static inline void philox_block(uint64_t seed, uint32_t offset, uint32_t i, uint32_t out[4])
{	/* counter (offset, 0, i, 0), key = seed lo/hi, 10 rounds: src/ccommon/rng_philox.c:26-48 */
	uint32_t cnt[4] = { offset, 0, i, 0 };
	uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed>>32) };
	for (unsigned r=0; r<10; ++r) {
		uint64_t v1 = (uint64_t)cnt[0] * philox_m[0];
		uint64_t v2 = (uint64_t)cnt[2] * philox_m[1];
		cnt[0] = (uint32_t)(v2>>32) ^ cnt[1] ^ key[0];
		cnt[1] = (uint32_t)v2;
		cnt[2] = (uint32_t)(v1>>32) ^ cnt[3] ^ key[1];
		cnt[3] = (uint32_t)v1;
		key[0] += philox_w[0];
		key[1] += philox_w[1];
	}
	memcpy(out, cnt, sizeof(cnt));
}

void orc_philox_raw(uint64_t seed, uint32_t offset, unsigned n, uint32_t* out2n)
{
	for (unsigned i=0;i<n;++i) { uint32_t c[4]; philox_block(seed, offset, i, c); out2n[2*i]=c[0]; out2n[2*i+1]=c[1]; }
}

void orc_rng_randn(OrcRng* S, unsigned n, float* out)
{	/* rng_philox_randn :23-51; box_muller :14-20 in double */
	const double two_pow32_inv = 2.3283064365386963e-10, two_pow32_inv_2pi = 1.4629180792671596e-09;
	#pragma omp parallel for schedule(static) if (n > 16384)
	for (unsigned i=0; i<n; ++i) {
		uint32_t c[4];
		philox_block(S->seed, S->offset, i, c);
		double u = ((double)c[0] + 0.5) * two_pow32_inv;
		double v = ((double)c[1] + 0.5) * two_pow32_inv_2pi;
		out[i] = sqrt(-2.0 * log(u)) * sin(v);
	}
	S->offset++;
}

/* ------------------------------------------------------------------ denoise */
OT* orc_unet_denoise_run(OParams* P, const char* prefix, const OrcUnetParams* U,
	const OT* x, const OT* cond, const OT* label, float sigma)
{	/* src/unet.c:460-498 */
	float t = orc_sigma_to_t(sigma);
	float c_in = 1 / sqrt(sigma*sigma + 1);
	OT *xs = ot_new(x->ne[0], x->ne[1], x->ne[2], x->ne[3]);
	int64_t n = ot_nel(x);
	for (int64_t i=0;i<n;++i) xs->d[i] = x->d[i] * c_in;
	OT *dx = orc_unet_graph(P, prefix, U, xs, t, cond, U->ch_adm_in ? label : NULL);
	ot_free(xs);
	if (U->vparam) {
		float c_skip = sigma / (sigma*sigma + 1), c_out = 1 / sqrt(sigma*sigma + 1);
		for (int64_t i=0;i<n;++i) dx->d[i] = dx->d[i]*c_out + x->d[i]*c_skip;
	}
	return dx;
}

static double now_s(void)
{
	struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec*1e-9;
}

int orc_generate_latent(OParams* P, const char* prefix, const OrcUnetParams* U,
	int lw, int lh, const OT* cond, const OT* label, const OT* uncond, const OT* unlabel,
	float cfg_scale, int n_step_req, float s_ancestral, uint64_t seed, int nfe_limit,
	float* latent_out, double* t_unet)
{
	float sigmas[1024];
	int n_step = orc_schedule(n_step_req, 1, 1, 0, sigmas);
	OrcRng rng = { seed, 0 };
	const int64_t n = (int64_t)lw*lh*U->n_ch_in;
	OT *x = ot_new(lw, lh, U->n_ch_in, 1);
	memset(x->d, 0, (size_t)n*sizeof(float));      /* mlimgsynth.c:1669-1670 */
	float *noise = (float*)malloc((size_t)n*sizeof(float));
	float solver_t = sigmas[0];                     /* sampling.c:92 */
	int nfe = 0;
	double tu = 0;
	for (int s=0; s<n_step; ++s) {                  /* dnsamp_step, sampling.c:119-185 */
		float s_up = 0, s_down = sigmas[s+1];
		if (s == 0) {
			orc_rng_randn(&rng, (unsigned)n, noise);
			for (int64_t i=0;i<n;++i) x->d[i] += noise[i] * sigmas[0];
		}
		if (s_ancestral > 0) orc_ancestral(sigmas[s], sigmas[s+1], s_ancestral, &s_down, &s_up);
		/* solver_euler_step, solvers.c:82-88 with mlis_denoise_dxdt, mlimgsynth.c:1572-1587 */
		float dt = s_down - solver_t;
		double t0 = now_s();
		OT *dx = orc_unet_denoise_run(P, prefix, U, x, cond, label, solver_t); nfe++;
		if (cfg_scale > 1 && (nfe_limit <= 0 || nfe < nfe_limit)) {
			OT *du = orc_unet_denoise_run(P, prefix, U, x, uncond, unlabel, solver_t); nfe++;
			float f = cfg_scale;
			for (int64_t i=0;i<n;++i) dx->d[i] = dx->d[i]*f + du->d[i]*(1-f);
			ot_free(du);
		}
		tu += now_s() - t0;
		for (int64_t i=0;i<n;++i) x->d[i] += dx->d[i] * dt;
		ot_free(dx);
		solver_t = s_down;
		if (s_up > 0 && s+1 != n_step) {
			orc_rng_randn(&rng, (unsigned)n, noise);
			for (int64_t i=0;i<n;++i) x->d[i] += noise[i] * s_up;
			solver_t = sigmas[s+1];
		}
		if (nfe_limit > 0 && nfe >= nfe_limit) break;
	}
	memcpy(latent_out, x->d, (size_t)n*sizeof(float));
	if (t_unet) *t_unet = tu;
	free(noise); ot_free(x);
	return nfe;
}

/* ------------------------------------------------------------------ general sampler
 * dnsamp_init / dnsamp_step (src/sampling.c:28-185) with the Solver classes of src/solvers.c:82-296 and
 * mlis_denoise_dxdt (src/mlimgsynth.c:1565-1587), restated in the reference's own structure: a solver owns
 * (t, i_step, tmp tensors) and calls dxdt(t, x). */
typedef struct {
	OParams *P; const char *prefix; const OrcUnetParams *U;
	const OT *cond, *label, *uncond, *unlabel;
	float cfg; int nfe; int64_t n; int lw, lh;
} DxCtx;

static void dxdt_eval(DxCtx* D, float t, const float* x, float* dx)
{
	OT *xt = ot_from(x, D->lw, D->lh, D->U->n_ch_in, 1);
	OT *d = orc_unet_denoise_run(D->P, D->prefix, D->U, xt, D->cond, D->label, t); D->nfe++;
	memcpy(dx, d->d, (size_t)D->n*sizeof(float));
	ot_free(d);
	float f = D->cfg;
	if (f > 1) {
		OT *u = orc_unet_denoise_run(D->P, D->prefix, D->U, xt, D->uncond, D->unlabel, t); D->nfe++;
		for (int64_t i=0;i<D->n;++i) dx[i] = dx[i]*f + u->d[i]*(1-f);
		ot_free(u);
	}
	ot_free(xt);
}

void orc_mask_downsize(const float* mask, int w, int h, int f, float* lmask)
{	/* ltensor_downsize(lmask, mask, f, f, 1, 1), src/localtensor.c:161-194 (mlis_mask_encode, mlimgsynth.c:1359-1365) */
	const int lw = w/f, lh = h/f;
	const float fn = 1.0f/(f*f);
	for (int i1=0;i1<lh;++i1) for (int i0=0;i0<lw;++i0) {
		float v = 0;
		for (int j1=0;j1<f;++j1) for (int j0=0;j0<f;++j0) v += mask[i0*f+j0 + (i1*f+j1)*w];
		lmask[i0 + i1*lw] = v * fn;
	}
}

int orc_sample_ex(OParams* P, const char* prefix, const OrcUnetParams* U, int lw, int lh,
	const OT* cond, const OT* label, const OT* uncond, const OT* unlabel, const OrcSampleOpts* O,
	uint64_t seed, uint32_t rng_offset, const float* init_latent, const float* lmask, float* latent_out)
{
	/* ---- dnsamp_init */
	int method = O->method > 0 ? O->method : 1;
	int nfe_solver = (method == 2 || method == 5) ? 2 : 1;
	int n_req = O->n_step < 1 ? 20 : O->n_step;
	if (nfe_solver > 1) n_req = (n_req + nfe_solver-1) / nfe_solver;
	float sigmas[1024];
	int n_step = orc_schedule(n_req, O->sched ? O->sched : 1, O->f_t_ini > 0 ? O->f_t_ini : 1, O->f_t_end, sigmas);
	OrcRng rng = { seed, rng_offset };
	const int64_t n = (int64_t)lw*lh*U->n_ch_in;
	DxCtx D = { P, prefix, U, cond, label, uncond, unlabel, O->cfg_scale, 0, n, lw, lh };
	float *x = (float*)calloc(n, sizeof(float)), *x0 = (float*)calloc(n, sizeof(float)), *noise = (float*)malloc(n*sizeof(float));
	float *dx = (float*)malloc(n*sizeof(float)), *tmp0 = (float*)calloc(n, sizeof(float)), *tmp1 = (float*)calloc(n, sizeof(float));
	if (init_latent) memcpy(x, init_latent, (size_t)n*sizeof(float));
	float S_t = sigmas[0];
	unsigned S_istep = 0;
	float dt_prev = 0, h_last = 0;
	const int hw = lw*lh;
	#define MASK_APPLY() do { if (lmask) for (int64_t i=0;i<n;++i) { float m = lmask[i % hw]; x[i] = x0[i]*m + x[i]*(1-m); } } while (0)
	#define NOISE_ADD(SIG) do { orc_rng_randn(&rng, (unsigned)n, noise); for (int64_t i=0;i<n;++i) x[i] += noise[i] * (SIG); } while (0)
	for (int s=0; s<n_step; ++s) {
		float s_up = 0, s_down = sigmas[s+1];
		if (s == 0) {
			if (lmask) memcpy(x0, x, (size_t)n*sizeof(float));
			NOISE_ADD(sigmas[0]);
			MASK_APPLY();
		}
		if (O->s_noise > 0 && s > 0) {
			float s_curr = sigmas[s], s_hat = s_curr * sqrt(2) * O->s_noise, s_noise = sqrt(s_hat*s_hat - s_curr*s_curr);
			NOISE_ADD(s_noise);
			MASK_APPLY();
			S_t = s_hat;
		}
		if (O->s_ancestral > 0) orc_ancestral(sigmas[s], sigmas[s+1], O->s_ancestral, &s_down, &s_up);
		/* ---- solver_step(S, s_down, x) */
		const float t = s_down;
		switch (method) {
		case 1: {   /* solver_euler_step :82-88 */
			float dt = t - S_t;
			dxdt_eval(&D, S_t, x, dx);
			for (int64_t i=0;i<n;++i) x[i] += dx[i] * dt;
		} break;
		case 2: {   /* solver_heun_step :96-117 */
			float dt = t - S_t;
			float *x1 = tmp0, *d1 = tmp1;
			dxdt_eval(&D, S_t, x, dx);
			for (int64_t i=0;i<n;++i) x1[i] = x[i] + dx[i] * dt;
			if (!(t > 0)) { for (int64_t i=0;i<n;++i) x[i] = x1[i]; }
			else {
				dxdt_eval(&D, t, x1, d1);
				for (int64_t i=0;i<n;++i) x[i] += (dx[i] + d1[i]) * 0.5 * dt;
			}
		} break;
		case 3: {   /* solver_taylor3_step :137-168 */
			float dt = t - S_t;
			float *dp1 = tmp0, *dp2 = tmp1;
			dxdt_eval(&D, S_t, x, dx);
			for (int64_t i=0;i<n;++i) x[i] += dx[i] * dt;
			float idtp = S_istep >= 1 ? 1 / dt_prev : 0, f2 = S_istep >= 1 ? dt*dt/2 : 0, f3 = S_istep >= 2 ? dt*dt*dt/6 : 0;
			for (int64_t i=0;i<n;++i) {
				float d2 = (dx[i] - dp1[i]) * idtp, d3 = (d2 - dp2[i]) * idtp;
				x[i] += d2 * f2 + d3 * f3;
				dp1[i] = dx[i]; dp2[i] = d2;
			}
			dt_prev = dt;
		} break;
		case 4: {   /* solver_dpmpp2m_step :207-233 */
			float *dprev = tmp0;
			float a = t / S_t, h = -log(a), c = h / (2*h_last);
			if (S_istep == 0 || !(t > 0)) c = 0;
			dxdt_eval(&D, S_t, x, dx);
			for (int64_t i=0;i<n;++i) {
				float d0 = x[i] - S_t * dx[i], d1 = dprev[i], d = (1+c) * d0 - c * d1;
				x[i] = a * x[i] + (1-a) * d;
				dprev[i] = d0;
			}
			h_last = h;
		} break;
		case 5: {   /* solver_dpmpp2s_step :264-289 */
			float *x1 = tmp0, *dx1 = tmp1;
			dxdt_eval(&D, S_t, x, dx);
			if (!(t > 0)) { float dt = t - S_t; for (int64_t i=0;i<n;++i) x[i] += dx[i] * dt; }
			else {
				float t1 = sqrt(t * S_t), dt1 = t1 - S_t, a = t / S_t;
				for (int64_t i=0;i<n;++i) x1[i] = x[i] + dx[i] * dt1;
				dxdt_eval(&D, t1, x1, dx1);
				for (int64_t i=0;i<n;++i) { float d = x1[i] - t1 * dx1[i]; x[i] = a * x[i] + (1-a) * d; }
			}
		} break;
		default: free(x); free(x0); free(noise); free(dx); free(tmp0); free(tmp1); return -1;
		}
		S_t = t; S_istep++;
		if (s_up > 0 && s+1 != n_step) { NOISE_ADD(s_up); S_t = sigmas[s+1]; }
		MASK_APPLY();
	}
	#undef MASK_APPLY
	#undef NOISE_ADD
	memcpy(latent_out, x, (size_t)n*sizeof(float));
	free(x); free(x0); free(noise); free(dx); free(tmp0); free(tmp1);
	return D.nfe;
}
