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
