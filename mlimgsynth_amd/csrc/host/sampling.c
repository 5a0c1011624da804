/* Host-side sampler arithmetic — re-creation of the reference's src/sampling.c:28-185 (schedule,
 * ancestral split), src/solvers.c:82-88 (Euler) and src/ccommon/rng_philox.c:9-51 (Philox4x32-10 +
 * Box-Muller imitating torch-CUDA randn).  All scalar / integer work: it stays on the host exactly
 * as in the reference; only the latent-sized vector updates run on the device (mlsd_sampler_update).
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>

/* ------------------------------------------------------------------ Philox (integer part bit-exact, KAT src/test_rng.c:11-24) */
static const uint32_t philox_m[2] = {0xD2511F53, 0xCD9E8D57};
static const uint32_t philox_w[2] = {0x9E3779B9, 0xBB67AE85};
static const double two_pow32_inv     = 2.3283064365386963e-10;   /*   1/2^32 */
static const double two_pow32_inv_2pi = 1.4629180792671596e-09;   /* 2pi/2^32 */

/* values [i0, i1) of the draw the generator stands at (counter (offset, 0, i, 0)); does not advance it */
MLB_API void rng_philox_randn_range(const RngPhilox* S, unsigned i0, unsigned i1, float* out)
{
	const uint32_t k0 = (uint32_t)S->seed, k1 = (uint32_t)(S->seed >> 32);
	out -= i0;
	for (unsigned i=i0; i<i1; ++i) {
		uint32_t c0 = S->offset, c1 = 0, c2 = i, c3 = 0, ka = k0, kb = k1;
		for (unsigned r=0; r<10; ++r) {
			const uint64_t v1 = (uint64_t)c0 * philox_m[0], v2 = (uint64_t)c2 * philox_m[1];
			const uint32_t n0 = (uint32_t)(v2 >> 32) ^ c1 ^ ka, n2 = (uint32_t)(v1 >> 32) ^ c3 ^ kb;
			c1 = (uint32_t)v2; c3 = (uint32_t)v1; c0 = n0; c2 = n2;
			ka += philox_w[0]; kb += philox_w[1];
		}
		const double u = ((double)c0 + 0.5) * two_pow32_inv;
		const double v = ((double)c1 + 0.5) * two_pow32_inv_2pi;
		out[i] = sqrt(-2.0 * log(u)) * sin(v);
	}
}

MLB_API void rng_philox_randn(RngPhilox* S, unsigned n, float* out)
{
	rng_philox_randn_range(S, 0, n, out);
	S->offset++;
}

/* ------------------------------------------------------------------ schedule (dnsamp_init, src/sampling.c:28-96; 1-NFE solver) */
MLB_API int dnsamp_schedule(const UnetParams* P, int n_step, int sched, float f_t_ini, float f_t_end, float* sigmas)
{
	if (n_step < 1) n_step = 20;
	if (!(f_t_ini > 0)) f_t_ini = 1;
	n_step = n_step * (f_t_ini - f_t_end) + 0.5;
	if (n_step < 1) n_step = 1;
	sigmas[n_step] = 0;
	float t_ini = (P->n_step_train - 1) * f_t_ini, t_end = (P->n_step_train - 1) * f_t_end;
	if (!sched) sched = DNSAMP_SCHED_UNIFORM;
	switch (sched) {
	case DNSAMP_SCHED_UNIFORM: {
		float b = t_ini, f = n_step > 1 ? (t_end - t_ini) / (n_step - 1) : 0;
		for (unsigned i=0; i<(unsigned)n_step; ++i) sigmas[i] = unet_t_to_sigma(P, b + i*f);
	} break;
	case DNSAMP_SCHED_KARRAS: {
		float smin = unet_t_to_sigma(P, t_end), smax = unet_t_to_sigma(P, t_ini), p = 7,
		      sminp = pow(smin, 1/p), smaxp = pow(smax, 1/p), b = smaxp,
		      f = n_step > 1 ? (sminp - smaxp) / (n_step - 1) : 0;
		for (unsigned i=0; i<(unsigned)n_step; ++i) sigmas[i] = pow(b + i*f, p);
	} break;
	default: return mlsd_set_error(-1, "invalid sampling scheduler %d", sched);
	}
	return n_step;
}

/* ancestral step split, src/sampling.c:153-166 (k_diffusion get_ancestral_step) */
MLB_API void dnsamp_ancestral(float s1, float s2, float eta, float* s_down, float* s_up)
{
	float up = sqrt((s2*s2) * (s1*s1 - s2*s2) / (s1*s1));
	up *= eta;
	if (up > s2) up = s2;
	*s_up = up;
	*s_down = sqrt(s2*s2 - up*up);
}
