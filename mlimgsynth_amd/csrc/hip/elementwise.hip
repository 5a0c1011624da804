// Small HBM-bound kernels around the UNet/VAE graphs: layout conversion at the NCHW fp32
// boundary (LocalTensor, reference src/localtensor.h:16-20), timestep embedding, CLIP token
// embedding gather, the on-device Euler(-ancestral)+CFG update, finite check, and the
// deterministic synthetic-parameter generator.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

__global__ void nchw_to_nhwc_f16_kernel(const float* __restrict__ src, int n_src, int C, int HW, _Float16* __restrict__ dst,
                                        int n_dst, int Cpad, const float* __restrict__ scale, float scale0, int mode)
{
    const long total = (long)n_dst * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW), pix = (int)(i % HW);
        const int ns = n % n_src;
        const float s = scale ? scale[ns] : scale0;
        _Float16* d = dst + i * Cpad;
        for (int c = 0; c < Cpad; ++c) {
            float v = 0.f;
            if (c < C) {
                v = src[((long)ns * C + c) * HW + pix];
                if (mode == 1) v = tanhf(v * (1.0f / 3.0f)) * 3.0f;
                if (mode == 2) v = __fsub_rn(__fmul_rn(v, 2.0f), 1.0f);        // sdvae_encoder_pre [0,1] -> [-1,1], src/vae.h:36-40
                v *= s;
            }
            d[c] = (_Float16)v;
        }
    }
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ src, long ld, int n, int C, int HW, float* __restrict__ dst,
                                        float mul, float add)
{
    const long total = (long)n * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), img = (int)(t / C);
        dst[i] = src[((long)img * HW + pix) * ld + c] * mul + add;
    }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, float max_period, _Float16* __restrict__ out)
{
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int s = i / half, j = i % half;
    // ggml_timestep_embedding / sd_timestep_embedding (src/mlimgsynth.c:1485-1499): cos first, then sin
    const float freq = expf(-logf(max_period) * (float)j / (float)half);
    const float arg = t[s] * freq;
    out[(long)s * dim + j] = (_Float16)cosf(arg);
    out[(long)s * dim + j + half] = (_Float16)sinf(arg);
}

__global__ void act_f32_to_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, size_t n, int act)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i];
        switch (act) {
        case MLSD_ACT_SILU: v = silu_f(v); break;
        case MLSD_ACT_GELU: v = gelu_tanh_f(v); break;
        case MLSD_ACT_GELU_QUICK: v = gelu_quick_f(v); break;
        case MLSD_ACT_RELU: v = fmaxf(v, 0.f); break;
        default: break;
        }
        y[i] = (_Float16)v;
    }
}

__global__ void clip_embed_kernel(const int32_t* __restrict__ tokens, int n, int T, int d, const _Float16* __restrict__ tok_w,
                                  const float* __restrict__ pos_w, float* __restrict__ out)
{
    const long total = (long)n * T * d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const long r = i / d;
        const int t = (int)(r % T);
        const int tok = tokens[r];
        out[i] = (float)tok_w[(long)tok * d + c] + pos_w[(long)t * d + c];
    }
}

__global__ void sampler_update_kernel(float* __restrict__ x, const float* __restrict__ eps, long ld, int B, int C, int HW, float cfg,
                                      const float* __restrict__ dt, const float* __restrict__ noise,
                                      const float* __restrict__ s_up)
{
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), b = (int)(t / C);
        float dx = eps[((long)b * HW + pix) * ld + c];
        if (cfg > 1.0f) {
            const float du = eps[((long)(b + B) * HW + pix) * ld + c];
            dx = __fadd_rn(__fmul_rn(dx, cfg), __fmul_rn(du, 1.0f - cfg));   // dx*f + tmp*(1-f), src/mlimgsynth.c:1583
        }
        float v = __fadd_rn(x[i], __fmul_rn(dx, dt[b]));                       // x += dx*dt, src/solvers.c:86
        if (noise) v = __fadd_rn(v, __fmul_rn(noise[i], s_up[b]));            // x += noise*sigma_up, src/sampling.c:115
        x[i] = v;
    }
}

__global__ void noise_add_kernel(float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ s, int B, long per)
{
    const long total = (long)B * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        x[i] = __fadd_rn(x[i], __fmul_rn(noise[i], s[i / per]));   // x += noise*sigma, src/sampling.c:115
}

__global__ void count_nonfinite_kernel(const float* __restrict__ x, size_t n, int32_t* __restrict__ count)
{
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        bad += !(fabsf(v) <= 3.4028234e38f);
    }
    if (bad) atomicAdd(count, bad);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

// see oracle/o_core.c orc_synth_fill: value_i = offset + (float)(s_i - 131070) * kf
__global__ void synth_fill_kernel(void* __restrict__ dst, int dtype, long n, uint64_t key, float offset, float kf, int layout,
                                  long p0, long p1, long p2, long p3, long p4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint64_t u = mix64(key + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
        const int s = (int)(u & 0xFFFF) + (int)((u >> 16) & 0xFFFF) + (int)((u >> 32) & 0xFFFF) + (int)(u >> 48);
        const float v = __fadd_rn(offset, __fmul_rn((float)(s - 131070), kf));
        long o = i;
        if (layout == 1) {
            // reference conv weight index i = k0 + K0*(k1 + K1*(cin + Cin*cout)); p0=K0 p1=K1 p2=Cin p3=Cout p4=Cin_pad
            const long k0 = i % p0; long t = i / p0;
            const long k1 = t % p1; t /= p1;
            const long ci = t % p2; const long co = t / p2;
            o = ((co * p1 + k1) * p0 + k0) * p4 + ci;     // [cout][kh][kw][cin_pad]
        } else if (layout == 2) {
            // GEGLU linear weight [n_in=p0, 2*d], d=p1: row j<d is value j, row d+j is gate j.
            const long k = i % p0, row = i / p0;
            const long j = row < p1 ? row : row - p1;
            const long nr = (j >> 5) * 64 + (row < p1 ? 0 : 32) + (j & 31);
            o = nr * p0 + k;
        } else if (layout == 3) {
            const long row = i, j = row < p1 ? row : row - p1;
            o = (j >> 5) * 64 + (row < p1 ? 0 : 32) + (j & 31);
        }
        if (dtype == 1) ((_Float16*)dst)[o] = (_Float16)v;
        else ((float*)dst)[o] = v;
    }
}

inline int nblocks(long n, int per = 256, int cap = 4096)
{
    long b = (n + per - 1) / per;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// V operand of the cross attention that ends its q projection (gemm_pp.hpp PP_EPI_XATTN): vt[b][n][key] = v[b Tk + key][n], keys Tk .. 95 zero.  Runs once per
// conditioning (the context's K / V projections are step-invariant).  One thread per (b, n, 8 keys): 16-byte stores; the strided reads hit L2 (616 x 2 N bytes per layer).
__global__ __launch_bounds__(256) void xattn_pack_vt_kernel(const _Float16* __restrict__ v, long ldv, int n_img, int Tk, int N, _Float16* __restrict__ vt)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)n_img * N * 12;
    if (idx >= total) return;
    const int kc = (int)(idx % 12);
    const long bn = idx / 12;
    const int n = (int)(bn % N), b = (int)(bn / N);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int key = kc * 8 + e;
        o[e] = key < Tk ? v[((long)b * Tk + key) * ldv + n] : (_Float16)0.f;
    }
    *reinterpret_cast<f16x8*>(vt + bn * 96 + kc * 8) = o;
}

}  // namespace

extern "C" {

MLSD_API int mlsd_xattn_pack_vt(const void* v, int64_t ldv, int n_img, int Tk, int N, void* vt, void* stream)
{
    if (!v || !vt || n_img <= 0 || Tk <= 0 || Tk > 96 || N <= 0 || ((uintptr_t)vt & 15)) return mlsd_set_error(-1, "mlsd_xattn_pack_vt: bad arguments");
    const long total = (long)n_img * N * 12;
    hipLaunchKernelGGL(xattn_pack_vt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)v, (long)ldv, n_img, Tk, N, (_Float16*)vt);
    return mlsd_check_launch("xattn_pack_vt");
}

MLSD_API int mlsd_nchw_to_nhwc_f16(const float* src, int n_src, int C, int HW, void* dst, int n_dst, int Cpad,
                                   const float* scale, float scale0, int mode, void* stream)
{
    hipLaunchKernelGGL(nchw_to_nhwc_f16_kernel, dim3(nblocks((long)n_dst * HW)), dim3(256), 0, (hipStream_t)stream, src, n_src, C,
                       HW, (_Float16*)dst, n_dst, Cpad, scale, scale0, mode);
    return mlsd_check_launch("nchw_to_nhwc_f16");
}

MLSD_API int mlsd_nhwc_to_nchw_f32(const float* src, int64_t ld, int n, int C, int HW, float* dst, float mul, float add,
                                   void* stream)
{
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(nblocks((long)n * C * HW)), dim3(256), 0, (hipStream_t)stream, src, (long)ld, n,
                       C, HW, dst, mul, add);
    return mlsd_check_launch("nhwc_to_nchw_f32");
}

MLSD_API int mlsd_timestep_embedding(const float* t, int n, int dim, float max_period, void* out16, void* stream)
{
    const int total = n * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, n, dim,
                       max_period, (_Float16*)out16);
    return mlsd_check_launch("timestep_embedding");
}

MLSD_API int mlsd_act_f32_to_f16(const float* x, void* y16, size_t n, int act, void* stream)
{
    hipLaunchKernelGGL(act_f32_to_f16_kernel, dim3(nblocks((long)n)), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y16, n, act);
    return mlsd_check_launch("act_f32_to_f16");
}

MLSD_API int mlsd_clip_embed(const int32_t* tokens, int n, int T, int d, const void* tok_w16, const float* pos_w,
                             float* out, void* stream)
{
    hipLaunchKernelGGL(clip_embed_kernel, dim3(nblocks((long)n * T * d)), dim3(256), 0, (hipStream_t)stream, tokens, n, T, d,
                       (const _Float16*)tok_w16, pos_w, out);
    return mlsd_check_launch("clip_embed");
}

MLSD_API int mlsd_sampler_update(float* x, const float* eps, int64_t ld, int B, int C, int HW, float cfg,
                                 const float* dt, const float* noise, const float* s_up, void* stream)
{
    hipLaunchKernelGGL(sampler_update_kernel, dim3(nblocks((long)B * C * HW)), dim3(256), 0, (hipStream_t)stream, x, eps, (long)ld, B,
                       C, HW, cfg, dt, noise, s_up);
    return mlsd_check_launch("sampler_update");
}

MLSD_API int mlsd_noise_add(float* x, const float* noise, const float* s, int B, int64_t per, void* stream)
{
    hipLaunchKernelGGL(noise_add_kernel, dim3(nblocks((long)B * per)), dim3(256), 0, (hipStream_t)stream, x, noise, s, B, (long)per);
    return mlsd_check_launch("noise_add");
}

MLSD_API int mlsd_count_nonfinite(const float* x, size_t n, int32_t* count, void* stream)
{
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3(nblocks((long)n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, count);
    return mlsd_check_launch("count_nonfinite");
}

MLSD_API int mlsd_synth_fill(void* dst, int dtype, int64_t n, uint64_t key, float offset, float kf, int layout,
                             int64_t p0, int64_t p1, int64_t p2, int64_t p3, int64_t p4, void* stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(synth_fill_kernel, dim3(nblocks((long)n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, dst, dtype, (long)n,
                       key, offset, kf, layout, (long)p0, (long)p1, (long)p2, (long)p3, (long)p4);
    return mlsd_check_launch("synth_fill");
}

}  // extern "C"
