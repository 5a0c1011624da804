// Latent-sized vector updates of the denoising loop, on the device (the latent never leaves HBM between steps).
// Each kernel restates ONE loop of the reference with the SAME fp32 / fp64 operation order (no contraction: explicit
// _rn intrinsics), so the device sampler is bit-identical to the host arithmetic of
//   src/solvers.c:82-296 (euler, heun, taylor3, dpmpp2m, dpmpp2s), src/sampling.c:98-117 (mask blend, noise add),
//   src/mlimgsynth.c:1565-1587 (CFG mix), src/unet.c:490-494 (v-parameterisation rescale), src/vae.c:203-229 (latent sample).
// All of it is HBM-bound elementwise work over <= 4 MB: negligible next to a UNet evaluation.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

inline unsigned nblk(long n) { long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
#define GRID_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// dx = mix(vparam(eps_c), vparam(eps_u)); eps NHWC [N][HW][ld] (cond rows 0..B-1, uncond B..2B-1), x / dx NCHW [B][C][HW]
__global__ void dxdt_cfg_kernel(const float* __restrict__ eps, long ld, const float* __restrict__ x, float* __restrict__ dx,
                                int B, int C, int HW, float cfg, int vparam, float c_out, float c_skip)
{
    const long total = (long)B * C * HW;
    GRID_LOOP(i, total) {
        const int pix = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), b = (int)(t / C);
        float d = eps[((long)b * HW + pix) * ld + c];
        const float xv = vparam ? x[i] : 0.f;
        if (vparam) d = __fadd_rn(__fmul_rn(d, c_out), __fmul_rn(xv, c_skip));            // unet.c:493
        if (cfg > 1.0f) {
            float u = eps[((long)(b + B) * HW + pix) * ld + c];
            if (vparam) u = __fadd_rn(__fmul_rn(u, c_out), __fmul_rn(xv, c_skip));
            d = __fadd_rn(__fmul_rn(d, cfg), __fmul_rn(u, 1.0f - cfg));                    // mlimgsynth.c:1583
        }
        dx[i] = d;
    }
}

// fused Euler(-ancestral) step with the CFG mix (the headline path: one launch per step):
//   dx = c*f + u*(1-f) (mlimgsynth.c:1583);  x += dx*dt (solvers.c:86);  x += noise*s_up (sampling.c:115)
__global__ void euler_cfg_kernel(float* __restrict__ x, const float* __restrict__ eps, long ld, int B, int C, int HW, float cfg,
                                 float dt, const float* __restrict__ noise, float s_up)
{
    const long total = (long)B * C * HW;
    GRID_LOOP(i, total) {
        const int pix = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), b = (int)(t / C);
        float dx = eps[((long)b * HW + pix) * ld + c];
        if (cfg > 1.0f) {
            const float du = eps[((long)(b + B) * HW + pix) * ld + c];
            dx = __fadd_rn(__fmul_rn(dx, cfg), __fmul_rn(du, 1.0f - cfg));
        }
        float v = __fadd_rn(x[i], __fmul_rn(dx, dt));
        if (noise) v = __fadd_rn(v, __fmul_rn(noise[i], s_up));
        x[i] = v;
    }
}

// out = x + d*dt   (euler step solvers.c:86 with out == x; heun / dpmpp2s predictor :105, :278 with out = x1)
__global__ void axpy_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ d, float dt, long n)
{
    GRID_LOOP(i, n) out[i] = __fadd_rn(x[i], __fmul_rn(d[i], dt));
}

// heun corrector, solvers.c:112-113:  x += (dx + d1) * 0.5 * dt   -- the 0.5 literal makes the product double
__global__ void heun_corr_kernel(float* __restrict__ x, const float* __restrict__ dx, const float* __restrict__ d1, float dt, long n)
{
    GRID_LOOP(i, n) {
        const double p = __dmul_rn(__dmul_rn((double)__fadd_rn(dx[i], d1[i]), 0.5), (double)dt);
        x[i] = (float)__dadd_rn((double)x[i], p);
    }
}

// taylor3, solvers.c:150-165
__global__ void taylor3_kernel(float* __restrict__ x, const float* __restrict__ dx, float* __restrict__ dp1, float* __restrict__ dp2,
                               float dt, float idtp, float f2, float f3, long n)
{
    GRID_LOOP(i, n) {
        const float d = dx[i];
        float xv = __fadd_rn(x[i], __fmul_rn(d, dt));
        const float d2 = __fmul_rn(__fsub_rn(d, dp1[i]), idtp);
        const float d3 = __fmul_rn(__fsub_rn(d2, dp2[i]), idtp);
        xv = __fadd_rn(xv, __fadd_rn(__fmul_rn(d2, f2), __fmul_rn(d3, f3)));
        x[i] = xv; dp1[i] = d; dp2[i] = d2;
    }
}

// dpmpp2m, solvers.c:222-229
__global__ void dpmpp2m_kernel(float* __restrict__ x, const float* __restrict__ dx, float* __restrict__ dprev, float t_cur, float a, float c, long n)
{
    GRID_LOOP(i, n) {
        const float d0 = __fsub_rn(x[i], __fmul_rn(t_cur, dx[i]));
        const float d1 = dprev[i];
        const float d = __fsub_rn(__fmul_rn(__fadd_rn(1.0f, c), d0), __fmul_rn(c, d1));
        x[i] = __fadd_rn(__fmul_rn(a, x[i]), __fmul_rn(__fsub_rn(1.0f, a), d));
        dprev[i] = d0;
    }
}

// dpmpp2s second half, solvers.c:281-284
__global__ void dpmpp2s_kernel(float* __restrict__ x, const float* __restrict__ x1, const float* __restrict__ dx1, float t1, float a, long n)
{
    GRID_LOOP(i, n) {
        const float d = __fsub_rn(x1[i], __fmul_rn(t1, dx1[i]));
        x[i] = __fadd_rn(__fmul_rn(a, x[i]), __fmul_rn(__fsub_rn(1.0f, a), d));
    }
}

// x += noise * sigma (scalar), sampling.c:115
__global__ void noise_add_s_kernel(float* __restrict__ x, const float* __restrict__ noise, float s, long n)
{
    GRID_LOOP(i, n) x[i] = __fadd_rn(x[i], __fmul_rn(noise[i], s));
}

// in-painting blend, sampling.c:98-110:  x = x0*m + x*(1-m), m [HW] shared by channels (and images)
__global__ void mask_apply_kernel(float* __restrict__ x, const float* __restrict__ x0, const float* __restrict__ m, int HW, long n)
{
    GRID_LOOP(i, n) {
        const float mv = m[i % HW];
        x[i] = __fadd_rn(__fmul_rn(x0[i], mv), __fmul_rn(x[i], __fsub_rn(1.0f, mv)));
    }
}

// sdvae_latent_sample, vae.c:203-229: moments NHWC fp32 [B][HW][ld] (channels 0..cz-1 mean, cz..2cz-1 logvar) ->
// latent NCHW [B][cz][HW] = (mean + exp(clamp(logvar,-30,20)*0.5) * rand) * scale_factor; rand == NULL: the mean (sdvae_latent_mean)
__global__ void latent_sample_kernel(const float* __restrict__ mom, long ld, const float* __restrict__ rnd, float* __restrict__ out,
                                     int B, int cz, int HW, float scale)
{
    const long total = (long)B * cz * HW;
    GRID_LOOP(i, total) {
        const int pix = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % cz), b = (int)(t / cz);
        const float* p = mom + ((long)b * HW + pix) * ld;
        float v = p[c];
        if (rnd) {
            float lv = p[cz + c];
            lv = lv < -30.f ? -30.f : (lv > 20.f ? 20.f : lv);
            // mean[i] + exp(clamped * 0.5) * rand[i]: double (0.5 literal, exp) -> float on store
            v = (float)__dadd_rn((double)v, __dmul_rn(exp(__dmul_rn((double)lv, 0.5)), (double)rnd[i]));
        }
        out[i] = __fmul_rn(v, scale);
    }
}

// the same for moments already in the reference layout NCHW [B][2cz][HW] (tiled encode pastes tiles into that layout)
__global__ void latent_sample_nchw_kernel(const float* __restrict__ mom, const float* __restrict__ rnd, float* __restrict__ out,
                                          int B, int cz, int HW, float scale)
{
    const long total = (long)B * cz * HW;
    GRID_LOOP(i, total) {
        const long per = (long)cz * HW;
        const int b = (int)(i / per);
        const long j = i - (long)b * per;
        float v = mom[(long)b * 2 * per + j];
        if (rnd) {
            float lv = mom[(long)b * 2 * per + per + j];
            lv = lv < -30.f ? -30.f : (lv > 20.f ? 20.f : lv);
            v = (float)__dadd_rn((double)v, __dmul_rn(exp(__dmul_rn((double)lv, 0.5)), (double)rnd[i]));
        }
        out[i] = __fmul_rn(v, scale);
    }
}

// ltensor_copy_slice2 (src/localtensor.h:84-94) on NCHW fp32 planes: dst[pl][di1+y][di0+x] = src[pl][si1+y][si0+x], n0 x n1 window
__global__ void copy_slice2_kernel(float* __restrict__ dst, int dw, int dh, const float* __restrict__ src, int sw, int sh,
                                   int n0, int n1, int di0, int di1, int si0, int si1, int planes)
{
    const long total = (long)planes * n1 * n0;
    GRID_LOOP(i, total) {
        const int x = (int)(i % n0);
        const long t = i / n0;
        const int y = (int)(t % n1), pl = (int)(t / n1);
        dst[((long)pl * dh + di1 + y) * dw + di0 + x] = src[((long)pl * sh + si1 + y) * sw + si0 + x];
    }
}

}  // namespace

extern "C" {

MLSD_API int mlsd_latent_sample_nchw(const float* moments, const float* rnd, float* latent, int B, int cz, int HW, float scale, void* stream)
{
    hipLaunchKernelGGL(latent_sample_nchw_kernel, dim3(nblk((long)B * cz * HW)), dim3(256), 0, (hipStream_t)stream, moments, rnd, latent,
                       B, cz, HW, scale);
    return mlsd_check_launch("latent_sample_nchw");
}

MLSD_API int mlsd_copy_slice2(float* dst, int dw, int dh, const float* src, int sw, int sh, int n0, int n1, int di0, int di1,
                              int si0, int si1, int planes, void* stream)
{
    if (n0 <= 0 || n1 <= 0 || planes <= 0) return 0;
    if (di0 < 0 || di1 < 0 || si0 < 0 || si1 < 0 || di0 + n0 > dw || di1 + n1 > dh || si0 + n0 > sw || si1 + n1 > sh)
        return mlsd_set_error(-1, "mlsd_copy_slice2: window out of bounds");
    hipLaunchKernelGGL(copy_slice2_kernel, dim3(nblk((long)planes * n0 * n1)), dim3(256), 0, (hipStream_t)stream, dst, dw, dh, src, sw, sh,
                       n0, n1, di0, di1, si0, si1, planes);
    return mlsd_check_launch("copy_slice2");
}


MLSD_API int mlsd_dxdt_cfg(const float* eps, int64_t ld, const float* x_eval, float* dx, int B, int C, int HW, float cfg,
                           int vparam, float c_out, float c_skip, void* stream)
{
    hipLaunchKernelGGL(dxdt_cfg_kernel, dim3(nblk((long)B * C * HW)), dim3(256), 0, (hipStream_t)stream, eps, (long)ld, x_eval, dx,
                       B, C, HW, cfg, vparam, c_out, c_skip);
    return mlsd_check_launch("dxdt_cfg");
}

MLSD_API int mlsd_euler_cfg_update(float* x, const float* eps, int64_t ld, int B, int C, int HW, float cfg, float dt,
                                   const float* noise, float s_up, void* stream)
{
    hipLaunchKernelGGL(euler_cfg_kernel, dim3(nblk((long)B * C * HW)), dim3(256), 0, (hipStream_t)stream, x, eps, (long)ld, B, C, HW,
                       cfg, dt, noise, s_up);
    return mlsd_check_launch("euler_cfg_update");
}

MLSD_API int mlsd_vec_axpy(float* out, const float* x, const float* d, float dt, int64_t n, void* stream)
{
    hipLaunchKernelGGL(axpy_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, out, x, d, dt, (long)n);
    return mlsd_check_launch("vec_axpy");
}

MLSD_API int mlsd_solver_heun_corr(float* x, const float* dx, const float* d1, float dt, int64_t n, void* stream)
{
    hipLaunchKernelGGL(heun_corr_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, dx, d1, dt, (long)n);
    return mlsd_check_launch("solver_heun_corr");
}

MLSD_API int mlsd_solver_taylor3(float* x, const float* dx, float* dp1, float* dp2, float dt, float idtp, float f2, float f3,
                                 int64_t n, void* stream)
{
    hipLaunchKernelGGL(taylor3_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, dx, dp1, dp2, dt, idtp, f2, f3, (long)n);
    return mlsd_check_launch("solver_taylor3");
}

MLSD_API int mlsd_solver_dpmpp2m(float* x, const float* dx, float* dprev, float t_cur, float a, float c, int64_t n, void* stream)
{
    hipLaunchKernelGGL(dpmpp2m_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, dx, dprev, t_cur, a, c, (long)n);
    return mlsd_check_launch("solver_dpmpp2m");
}

MLSD_API int mlsd_solver_dpmpp2s(float* x, const float* x1, const float* dx1, float t1, float a, int64_t n, void* stream)
{
    hipLaunchKernelGGL(dpmpp2s_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, x1, dx1, t1, a, (long)n);
    return mlsd_check_launch("solver_dpmpp2s");
}

/* two constants into two short device vectors, from kernel arguments: the per-evaluation scalars of the UNet (timestep for every row, c_in for every image) reach
 * the device without a copy-engine job (a 64-byte hipMemcpyAsync queues behind whatever that engine is moving: with streamed weights, a 500 MB upload) */
__global__ void fill2_kernel(float* a, int na, float va, float* b, int nb, float vb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = va;
    if (i < nb) b[i] = vb;
}
MLSD_API int mlsd_fill2_f32(float* a, int na, float va, float* b, int nb, float vb, void* stream)
{
    const int n = na > nb ? na : nb;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill2_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, na, va, b, nb, vb);
    return mlsd_check_launch("fill2_kernel");
}

MLSD_API int mlsd_noise_add_s(float* x, const float* noise, float s, int64_t n, void* stream)
{
    hipLaunchKernelGGL(noise_add_s_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, noise, s, (long)n);
    return mlsd_check_launch("noise_add_s");
}

MLSD_API int mlsd_mask_apply(float* x, const float* x0, const float* mask, int HW, int64_t n, void* stream)
{
    hipLaunchKernelGGL(mask_apply_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, x0, mask, HW, (long)n);
    return mlsd_check_launch("mask_apply");
}

MLSD_API int mlsd_latent_sample(const float* moments, int64_t ld, const float* rnd, float* latent, int B, int cz, int HW,
                                float scale, void* stream)
{
    hipLaunchKernelGGL(latent_sample_kernel, dim3(nblk((long)B * cz * HW)), dim3(256), 0, (hipStream_t)stream, moments, (long)ld, rnd,
                       latent, B, cz, HW, scale);
    return mlsd_check_launch("latent_sample");
}

}  // extern "C"
