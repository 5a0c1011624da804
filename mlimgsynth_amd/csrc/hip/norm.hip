// GroupNorm(+affine+SiLU) and LayerNorm(+affine) for gfx950: HBM-bound wavefront reductions.
//
// GroupNorm (reference: ggml_group_norm + mul + add [+ silu], src/mlblock_nn.c:78-103,135-136):
//   input  fp32 channels-last, optionally TWO sources concatenated along channels (the
//          ggml_concat of src/unet.c:233 never materialises),
//   pass 1 (gn_stats):  per (image, pixel-chunk) partial sums of (x-K) and (x-K)^2 per group,
//          K = the group's first element (shifted sums: no catastrophic cancellation),
//          combined deterministically (no atomics),
//   pass 2 (gn_apply):  every block re-derives mean/rstd of its image from the partials in
//          double, builds per-channel scale/shift in LDS, then streams x -> fp16 y
//          (normalise, affine, SiLU) and optionally a raw fp16 copy for the 1x1 skip conv.
//   Traffic: 2 fp32 reads + 1 fp16 write per element (the statistic is a global reduction, so
//   one re-read is the minimum without fusing the statistics into the producer's epilogue).
//
// LayerNorm (reference: ggml_norm + mul + add, src/mlblock_nn.c:58-75): one wavefront per row,
//   row held in registers, two-pass mean/variance, fp16 (and/or fp32) output.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

constexpr int GN_MAXG = 32;
constexpr int GN_MAXQ = 3;    // channel quads per thread: supports C up to 4*256*3 = 3072

struct GnP {
    const float *x1, *x2;
    long ld1, ld2;
    int C1, C2, C, HW, G, cg;
    int nchunk, pix_per_chunk;
    float eps;
    const float *gamma, *beta;
    int silu;
    _Float16* y16;
    _Float16* raw16;
    float* ws;  // [n_img][nchunk][G][2]
    // producer statistics (column sums / sums of squares per block of rb rows, written by the GEMM epilogues): when set, pass 1
    // is gn_finalize over these few KB..MB instead of gn_stats over the whole fp32 map
    const float *cs1, *cs2;
    int rb1, rb2;
    const float* mr;  // [n_img][G][2] mean, rstd (gn_finalize -> gn_apply), or null
};

__device__ __forceinline__ const float* gn_src(const GnP& p, int img, int pix, int c)
{
    return c < p.C1 ? p.x1 + ((long)img * p.HW + pix) * p.ld1 + c : p.x2 + ((long)img * p.HW + pix) * p.ld2 + (c - p.C1);
}

// Thread mapping shared by both passes: a thread owns fixed channel quads (16-byte loads) and walks the
// pixels of its chunk with a fixed stride, so there is no per-element index arithmetic:
//   Q = C/4 quads per pixel.  Q >= 256: every thread owns quads tid, tid+256, .. and visits every pixel.
//                             Q <  256: PL = 256/Q pixel lanes; thread = (lane pl, quad q), pixel stride PL.
struct GnMap { int PL, NQ, pl, q0; bool active; };
__device__ __forceinline__ GnMap gn_map(int Q, int tid)
{
    GnMap m;
    if (Q >= 256) { m.PL = 1; m.NQ = (Q + 255) / 256; m.pl = 0; m.q0 = tid; m.active = true; }
    else { m.PL = 256 / Q; m.NQ = 1; m.pl = tid / Q; m.q0 = tid - m.pl * Q; m.active = m.pl < m.PL; }
    return m;
}

__global__ __launch_bounds__(256) void gn_stats(const GnP p)
{
    // per (thread, quad, channel pair): shifted sum and sum of squares.  A quad may straddle two groups
    // (channels per group is even but not always a multiple of 4), a pair never does.
    __shared__ float s1[256 * GN_MAXQ * 2], s2[256 * GN_MAXQ * 2];
    const int img = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int Q = p.C >> 2;
    const GnMap m = gn_map(Q, tid);
    const int pix0 = chunk * p.pix_per_chunk;
    const int pix1 = min(pix0 + p.pix_per_chunk, p.HW);

    float a1[GN_MAXQ][2], a2[GN_MAXQ][2], K[GN_MAXQ][2];
    const float* base[GN_MAXQ];
    long ld[GN_MAXQ];
#pragma unroll
    for (int i = 0; i < GN_MAXQ; ++i) {
        a1[i][0] = a1[i][1] = a2[i][0] = a2[i][1] = 0.f;
        K[i][0] = K[i][1] = 0.f;
        base[i] = nullptr; ld[i] = 0;
        const int q = m.q0 + i * 256;
        if (i < m.NQ && q < Q) {
            const int c = 4 * q;
            K[i][0] = *gn_src(p, img, 0, (c / p.cg) * p.cg);
            K[i][1] = *gn_src(p, img, 0, ((c + 2) / p.cg) * p.cg);
            base[i] = gn_src(p, img, 0, c);
            ld[i] = c < p.C1 ? p.ld1 : p.ld2;
        }
    }
    if (m.active) {
        // 4 pixels per trip: 4*NQ independent 16-byte loads in flight per thread (HBM latency, not issue, bounds this pass)
        for (int pix = pix0 + m.pl; pix < pix1; pix += 4 * m.PL) {
#pragma unroll
            for (int i = 0; i < GN_MAXQ; ++i) {
                if (base[i]) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int px = pix + u * m.PL;
                        v[u] = px < pix1 ? *reinterpret_cast<const float4*>(base[i] + (long)px * ld[i])
                                         : make_float4(K[i][0], K[i][0], K[i][1], K[i][1]);   // contributes exactly 0
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d0 = v[u].x - K[i][0], d1 = v[u].y - K[i][0], d2 = v[u].z - K[i][1], d3 = v[u].w - K[i][1];
                        a1[i][0] += d0 + d1; a2[i][0] += d0 * d0 + d1 * d1;
                        a1[i][1] += d2 + d3; a2[i][1] += d2 * d2 + d3 * d3;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < GN_MAXQ; ++i) {
        s1[(i * 256 + tid) * 2] = a1[i][0]; s1[(i * 256 + tid) * 2 + 1] = a1[i][1];
        s2[(i * 256 + tid) * 2] = a2[i][0]; s2[(i * 256 + tid) * 2 + 1] = a2[i][1];
    }
    __syncthreads();
    if (tid < p.G) {
        // deterministic combine: channel pairs [g*cg/2, (g+1)*cg/2) of group g, over all pixel lanes
        const int g = tid, pr0 = g * (p.cg >> 1), pr1 = pr0 + (p.cg >> 1);
        float t1 = 0.f, t2 = 0.f;
        for (int pr = pr0; pr < pr1; ++pr) {
            const int q = pr >> 1, half = pr & 1;
            if (Q >= 256) {
                const int i = q >> 8, t = q & 255;
                t1 += s1[(i * 256 + t) * 2 + half]; t2 += s2[(i * 256 + t) * 2 + half];
            } else {
                for (int l = 0; l < m.PL; ++l) { t1 += s1[(l * Q + q) * 2 + half]; t2 += s2[(l * Q + q) * 2 + half]; }
            }
        }
        float* w = p.ws + (((long)img * p.nchunk + chunk) * p.G + g) * 2;
        w[0] = t1; w[1] = t2;
    }
}

__global__ __launch_bounds__(256) void gn_apply(const GnP p)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];  // scale[C], shift[C]
    __shared__ float g_mean[GN_MAXG], g_rstd[GN_MAXG];
    __shared__ double r1[8][GN_MAXG], r2[8][GN_MAXG];
    float* sc = sm;
    float* sh = sm + p.C;
    const int img = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    if (p.mr) {
        if (tid < p.G) { g_mean[tid] = p.mr[((long)img * p.G + tid) * 2]; g_rstd[tid] = p.mr[((long)img * p.G + tid) * 2 + 1]; }
    } else {
        // partial sums of the image's chunks: 8 stripes of chunks per group in parallel, then a fixed-order
        // combine (deterministic; a serial walk over up to 256 chunks was most of this kernel at batch 1)
        const int g = tid & 31, st = tid >> 5;
        double t1 = 0, t2 = 0;
        if (g < p.G) {
            // 4 independent 8-byte loads in flight per trip: a one-load-per-trip loop pays an L2 round trip per chunk
            // (up to 32 of them per thread: most of this kernel's time on small maps)
            const float2* w = reinterpret_cast<const float2*>(p.ws) + (long)img * p.nchunk * p.G + g;
            int c = st;
            for (; c + 24 < p.nchunk; c += 32) {
                const float2 a = w[(long)c * p.G], b = w[(long)(c + 8) * p.G], d = w[(long)(c + 16) * p.G], e = w[(long)(c + 24) * p.G];
                t1 += ((double)a.x + (double)b.x) + ((double)d.x + (double)e.x);
                t2 += ((double)a.y + (double)b.y) + ((double)d.y + (double)e.y);
            }
            for (; c < p.nchunk; c += 8) { const float2 a = w[(long)c * p.G]; t1 += a.x; t2 += a.y; }
        }
        r1[st][g] = t1; r2[st][g] = t2;
    }
    __syncthreads();
    if (!p.mr && tid < p.G) {
        double t1 = 0, t2 = 0;
#pragma unroll
        for (int st = 0; st < 8; ++st) { t1 += r1[st][tid]; t2 += r2[st][tid]; }
        const double cnt = (double)p.cg * p.HW;
        const double K = *gn_src(p, img, 0, tid * p.cg);
        const double mu = t1 / cnt;
        double var = t2 / cnt - mu * mu;
        if (var < 0) var = 0;
        g_mean[tid] = (float)(K + mu);
        g_rstd[tid] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g = c / p.cg;
        const float s = g_rstd[g] * p.gamma[c];
        sc[c] = s;
        sh[c] = p.beta[c] - g_mean[g] * s;
    }
    __syncthreads();
    const int Q = p.C >> 2;
    const GnMap m = gn_map(Q, tid);
    const int pix0 = chunk * p.pix_per_chunk;
    const int pix1 = min(pix0 + p.pix_per_chunk, p.HW);
    float4 s4[GN_MAXQ], b4[GN_MAXQ];
    const float* base[GN_MAXQ];
    long ld[GN_MAXQ];
#pragma unroll
    for (int i = 0; i < GN_MAXQ; ++i) {
        base[i] = nullptr; ld[i] = 0;
        s4[i] = b4[i] = make_float4(0, 0, 0, 0);
        const int q = m.q0 + i * 256;
        if (i < m.NQ && q < Q) {
            const int c = 4 * q;
            s4[i] = *reinterpret_cast<const float4*>(sc + c);
            b4[i] = *reinterpret_cast<const float4*>(sh + c);
            base[i] = gn_src(p, img, 0, c);
            ld[i] = c < p.C1 ? p.ld1 : p.ld2;
        }
    }
    if (!m.active) return;
    for (int pix = pix0 + m.pl; pix < pix1; pix += 4 * m.PL) {
#pragma unroll
        for (int i = 0; i < GN_MAXQ; ++i) {
            if (base[i]) {
                float4 vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = pix + u * m.PL;
                    vv[u] = px < pix1 ? *reinterpret_cast<const float4*>(base[i] + (long)px * ld[i]) : make_float4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = pix + u * m.PL;
                    if (px >= pix1) continue;
                    const float4 v = vv[u];
                    float y0 = v.x * s4[i].x + b4[i].x, y1 = v.y * s4[i].y + b4[i].y, y2 = v.z * s4[i].z + b4[i].z, y3 = v.w * s4[i].w + b4[i].w;
                    if (p.silu) { y0 = silu_f(y0); y1 = silu_f(y1); y2 = silu_f(y2); y3 = silu_f(y3); }
                    const long o = ((long)img * p.HW + px) * p.C + 4 * (m.q0 + i * 256);
                    f16x4 h = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
                    *reinterpret_cast<f16x4*>(p.y16 + o) = h;
                    if (p.raw16) {
                        f16x4 r = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                        *reinterpret_cast<f16x4*>(p.raw16 + o) = r;
                    }
                }
            }
        }
    }
}

// Mean / rstd of one (image, group) from the producers' column statistics: sums over the image's row blocks and the
// group's channels, in double, fixed order (thread t takes items t, t+256, ..; then a fixed tree).  Unshifted sums: the
// partials are fp32 sums of 64..128 values, the cancellation in E[x^2] - mean^2 happens in double.
__global__ __launch_bounds__(256) void gn_finalize(const GnP p, float* __restrict__ mr)
{
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    double s = 0, q = 0;
    const int c0 = g * p.cg, c1 = c0 + p.cg;
    // the group's channels may straddle the two sources of a virtual concat
    for (int src = 0; src < 2; ++src) {
        const int lo = src ? max(c0, p.C1) : c0, hi = src ? c1 : min(c1, p.C1);
        if (lo >= hi) continue;
        const float* cs = src ? p.cs2 : p.cs1;
        const int Ci = src ? p.C2 : p.C1, rb = src ? p.rb2 : p.rb1, off = src ? p.C1 : 0;
        const int nrb = p.HW / rb, ncc = hi - lo;
        const float* base = cs + (long)img * nrb * 2 * Ci + (lo - off);
        for (int idx = tid; idx < nrb * ncc; idx += 256) {
            const int k = idx / ncc, cc = idx - k * ncc;
            const float* e = base + (long)k * 2 * Ci + cc;
            s += (double)e[0]; q += (double)e[Ci];
        }
    }
    rs[tid] = s; rq[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { rs[tid] += rs[tid + w]; rq[tid] += rq[tid + w]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)p.cg * p.HW;
        const double mu = rs[0] / cnt;
        double var = rq[0] / cnt - mu * mu;
        if (var < 0) var = 0;
        mr[((long)img * p.G + g) * 2] = (float)mu;
        mr[((long)img * p.G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

// Two-level form of gn_finalize for LARGE maps (round 4; the VAE's 512^2 / 1024^2 levels: 4 096 .. 16 384 row blocks per image).  gn_finalize gives one block per
// (image, group), and a group's cg columns are 16..64 bytes out of every 2 C floats of the statistics: 128 blocks reading 16-byte pieces at a 1 KB stride took 200-340 us
// for 67 MB.  Level 1: a block takes a contiguous range of row blocks for ALL columns (thread = column: consecutive lanes read consecutive floats), sums in double, then
// the group's columns through LDS (fixed order) -> part[image][chunk][group][2].  Level 2: one block per image adds the chunks in order and writes mean / rstd.
constexpr int GNF_MAXC = 2048;
__global__ __launch_bounds__(256) void gn_finalize_l1(const GnP p, double* __restrict__ part, int nch)
{
    __shared__ double cs_[GNF_MAXC], cq_[GNF_MAXC];
    const int chunk = blockIdx.x, img = blockIdx.y;
    for (int c = threadIdx.x; c < p.C; c += 256) {
        const int src = c >= p.C1;
        const float* cs = src ? p.cs2 : p.cs1;
        const int Ci = src ? p.C2 : p.C1, rb = src ? p.rb2 : p.rb1, cc = c - (src ? p.C1 : 0);
        const int nrb = p.HW / rb;
        const int k0 = (int)((long)chunk * nrb / nch), k1 = (int)((long)(chunk + 1) * nrb / nch);
        const float* e = cs + ((long)img * nrb + k0) * 2 * Ci + cc;
        double s = 0, q = 0;
        for (int k = k0; k < k1; ++k, e += 2 * Ci) { s += (double)e[0]; q += (double)e[Ci]; }
        cs_[c] = s; cq_[c] = q;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < p.G; g += 256) {
        double s = 0, q = 0;
        for (int j = 0; j < p.cg; ++j) { s += cs_[g * p.cg + j]; q += cq_[g * p.cg + j]; }
        double* o = part + (((long)img * nch + chunk) * p.G + g) * 2;
        o[0] = s; o[1] = q;
    }
}
__global__ __launch_bounds__(256) void gn_finalize_l2(const GnP p, const double* __restrict__ part, int nch, float* __restrict__ mr)
{
    const int img = blockIdx.x;
    for (int g = threadIdx.x; g < p.G; g += 256) {
        double s = 0, q = 0;
        for (int c = 0; c < nch; ++c) { const double* e = part + (((long)img * nch + c) * p.G + g) * 2; s += e[0]; q += e[1]; }
        const double cnt = (double)p.cg * p.HW;
        const double mu = s / cnt;
        double var = q / cnt - mu * mu;
        if (var < 0) var = 0;
        mr[((long)img * p.G + g) * 2] = (float)mu;
        mr[((long)img * p.G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

// ONE-DISPATCH GroupNorm for small maps (round 4): SD1.5 batch 1 is bound by its dispatch count (~4.5 us per dependent dispatch, 498 per evaluation), and on
// the 8x8 .. 32x32 levels an (image, group) slab is 10..80 KB.  One block per (image, group) holds its slab in registers (each element read ONCE), takes the mean
// and the centred sum of squares with two block reductions (fixed order: deterministic, and better conditioned than single-pass sums), normalises and stores.
// A slab row is cg*4 bytes of one pixel (160 B at C = 1280), so this form is for cg % 4 == 0 and slabs of <= GN1_MAXI float4 per thread (40 KB); the two-kernel form
// keeps everything else (its 1024 blocks stream the big maps better than 64 blocks can: measured in round 2).
constexpr int GN1_MAXI = 10;      // float4 per thread: slabs of up to 256 * 10 * 4 = 10240 elements (40 KB).  Measured per GroupNorm in the SD1.5 b1 plan (rocprofv3 dispatch times,
// two kernels -> one): n2 hw64 c1280 11.2 -> 7.4 us, hw64 c2560 16.2 -> 8.5, hw256 c1280 13.8 -> 12.4; 80 KB slabs LOSE (hw1024 c640 13.5 -> 19.5 us: 64 blocks cannot pull 5 MB as fast as 1024 can)
__global__ __launch_bounds__(256) void gn_small_kernel(const GnP p)
{
    __shared__ float red[8];
    const int g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Q4 = p.cg >> 2, items = p.HW * Q4, c0 = g * p.cg;
    float4 v[GN1_MAXI];
    // all loads first, unconditionally (clamped item index: a branch per load would serialise them behind one s_waitcnt each -- the first version of this
    // kernel took 13 us for an 80 KB slab that way); NI = the trips this slab needs (block-uniform)
    const int NI = (items + 255) >> 8;
#pragma unroll
    for (int i = 0; i < GN1_MAXI; ++i) {
        if (i < NI) {
            const int it = min(tid + i * 256, items - 1);
            const int pix = it / Q4, c = c0 + 4 * (it - pix * Q4);
            v[i] = *reinterpret_cast<const float4*>(gn_src(p, img, pix, c));
        } else v[i] = make_float4(0, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GN1_MAXI; ++i) {
        const bool in = tid + i * 256 < items;
        s += in ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float cnt = (float)p.cg * (float)p.HW;
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GN1_MAXI; ++i) {
        if (tid + i * 256 < items) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[4 + wave] = s2;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / cnt + p.eps);
#pragma unroll
    for (int i = 0; i < GN1_MAXI; ++i) {
        const int it = tid + i * 256;
        if (it < items) {
            const int pix = it / Q4, c = c0 + 4 * (it - pix * Q4);
            const float4 ga = *reinterpret_cast<const float4*>(p.gamma + c), be = *reinterpret_cast<const float4*>(p.beta + c);
            float y0 = (v[i].x - mean) * rstd * ga.x + be.x, y1 = (v[i].y - mean) * rstd * ga.y + be.y;
            float y2 = (v[i].z - mean) * rstd * ga.z + be.z, y3 = (v[i].w - mean) * rstd * ga.w + be.w;
            if (p.silu) { y0 = silu_f(y0); y1 = silu_f(y1); y2 = silu_f(y2); y3 = silu_f(y3); }
            const long o = ((long)img * p.HW + pix) * p.C + c;
            f16x4 h = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
            *reinterpret_cast<f16x4*>(p.y16 + o) = h;
            if (p.raw16) {
                f16x4 r = {(_Float16)v[i].x, (_Float16)v[i].y, (_Float16)v[i].z, (_Float16)v[i].w};
                *reinterpret_cast<f16x4*>(p.raw16 + o) = r;
            }
        }
    }
}

int g_gn_single = -1;     // -1: not decided (environment MLSD_GN_SINGLE, default on), 0 / 1
bool gn_single_ok(int n_img, int HW, int C, int n_grp)
{
    if (g_gn_single < 0) { const char* e = getenv("MLSD_GN_SINGLE"); g_gn_single = (e && *e == '0') ? 0 : 1; }
    if (!g_gn_single || n_grp <= 0 || C % n_grp) return false;
    const int cg = C / n_grp;
    // few images only: with n_img * n_grp >= 256 blocks of the two-kernel form already fill the chip, and big batches are not dispatch-bound
    return !(cg & 3) && (long)HW * (cg >> 2) <= 256L * GN1_MAXI && n_img * n_grp <= 128;
}

// pixel chunks per image: enough blocks (n_img * chunks ~ 1024) to occupy 256 CUs at batch 1, down to ONE pixel per block
// on the 8x8 / 16x16 maps of the UNet's inner levels (a block walks its pixels serially, one memory round trip per
// 4-pixel trip: with 16-pixel chunks those maps ran 8 blocks x 4 dependent trips)
int gn_chunks(int HW, int n_img)
{
    int cap = 1024 / (n_img > 0 ? n_img : 1);
    if (cap < 64) cap = 64;
    if (cap > 256) cap = 256;
    int c = HW < cap ? HW : cap;
    if (c < 1) c = 1;
    return c;
}

// ---------------------------------------------------------------- LayerNorm
constexpr int LN_MAXQ = 8;  // float4 per lane: d up to 64*4*8 = 2048

__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, long ldx, int rows, int d, float eps,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 _Float16* __restrict__ y16, float* __restrict__ y32)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int Q = d >> 2;
    const float* xr = x + (long)row * ldx;
    float4 v[LN_MAXQ];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
        const int q = lane + i * 64;
        if (q < Q) { v[i] = *reinterpret_cast<const float4*>(xr + q * 4); s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        else v[i] = make_float4(0, 0, 0, 0);
    }
    const float mean = wave_sum(s) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
        const int q = lane + i * 64;
        if (q < Q) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
        const int q = lane + i * 64;
        if (q < Q) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + q * 4);
            const float4 b = beta ? *reinterpret_cast<const float4*>(beta + q * 4) : make_float4(0, 0, 0, 0);
            const float y0 = v[i].x * rstd * g.x + b.x, y1 = v[i].y * rstd * g.y + b.y;
            const float y2 = v[i].z * rstd * g.z + b.z, y3 = v[i].w * rstd * g.w + b.w;
            if (y16) {
                f16x4 h = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
                *reinterpret_cast<f16x4*>(y16 + (long)row * d + q * 4) = h;
            }
            if (y32) *reinterpret_cast<float4*>(y32 + (long)row * d + q * 4) = make_float4(y0, y1, y2, y3);
        }
    }
}

// Streaming form for many rows: a wave walks rows w, w + W, w + 2W, ... and fetches its next row before it reduces the current
// one.  With one row per wave (ln_kernel at 8192 rows = exactly one resident round) the whole chip first reads, then reduces,
// then writes, in step; here reads of row i+1 overlap the statistics and the stores of row i.  NQ = float4 per lane.
int g_ln_stream_blocks = 1024;  // blocks of 4 waves (4 per CU): measured best of 256..2048 at 8192 x 1280 and 32768 x 640

template <int NQ>
__global__ __launch_bounds__(256) void ln_stream_kernel(const float* __restrict__ x, long ldx, int rows, int d, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        _Float16* __restrict__ y16, float* __restrict__ y32)
{
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Q = d >> 2;
    float4 g[NQ], b[NQ], cur[NQ], nxt[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = lane + i * 64;
        const bool in = q < Q;
        g[i] = in ? *reinterpret_cast<const float4*>(gamma + q * 4) : make_float4(0, 0, 0, 0);
        b[i] = (in && beta) ? *reinterpret_cast<const float4*>(beta + q * 4) : make_float4(0, 0, 0, 0);
        cur[i] = in ? *reinterpret_cast<const float4*>(x + (long)row * ldx + q * 4) : make_float4(0, 0, 0, 0);
    }
    for (; row < rows; row += nw) {
        const int nrow = row + nw;
        if (nrow < rows) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int q = lane + i * 64;
                nxt[i] = q < Q ? *reinterpret_cast<const float4*>(x + (long)nrow * ldx + q * 4) : make_float4(0, 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) s += (cur[i].x + cur[i].y) + (cur[i].z + cur[i].w);
        const float mean = wave_sum(s) / (float)d;            // (the same operations as ln_kernel: bit-identical results)
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (lane + i * 64 < Q) {
                cur[i].x -= mean; cur[i].y -= mean; cur[i].z -= mean; cur[i].w -= mean;
                s2 += (cur[i].x * cur[i].x + cur[i].y * cur[i].y) + (cur[i].z * cur[i].z + cur[i].w * cur[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lane + i * 64;
            if (q < Q) {
                const float y0 = cur[i].x * rstd * g[i].x + b[i].x, y1 = cur[i].y * rstd * g[i].y + b[i].y;
                const float y2 = cur[i].z * rstd * g[i].z + b[i].z, y3 = cur[i].w * rstd * g[i].w + b[i].w;
                if (y16) {
                    f16x4 h = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
                    *reinterpret_cast<f16x4*>(y16 + (long)row * d + q * 4) = h;
                }
                if (y32) *reinterpret_cast<float4*>(y32 + (long)row * d + q * 4) = make_float4(y0, y1, y2, y3);
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) cur[i] = nxt[i];
    }
}

// ---------------------------------------------------------------- row softmax (VAE mid attention)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, long ld_in, _Float16* __restrict__ out,
                                                           long ld_out, int rows, int cols, float scale)
{
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = in + (long)row * ld_in;
    float mx = -3.0e38f;
    for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, xr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < cols; c += 256) s += __expf((xr[c] - mx) * scale);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    _Float16* yr = out + (long)row * ld_out;
    for (int c = tid; c < cols; c += 256) yr[c] = (_Float16)(__expf((xr[c] - mx) * scale) * inv);
}

// rows of up to 16384 columns (the VAE mid block at 1024^2: 16384 keys) are held in registers: ONE fp32 read and one fp16
// write per score instead of three reads (the materialised score matrix is 1 GiB per image, so this pass is HBM-bound)
constexpr int SM_MAXQ = 16;   // float4 per thread
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ in, long ld_in, _Float16* __restrict__ out,
                                                               long ld_out, int rows, int cols, float scale)
{
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = in + (long)row * ld_in;
    const int Q = cols >> 2;
    float4 v[SM_MAXQ];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < SM_MAXQ; ++i) {
        const int q = tid + i * 256;
        if (q < Q) {
            v[i] = *reinterpret_cast<const float4*>(xr + 4 * q);
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        } else v[i] = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXQ; ++i) {
        // same expression as the three-pass kernel: exp((x - max) * scale); lanes past the row hold -3e38 -> exp = 0
        v[i].x = __expf((v[i].x - mx) * scale); v[i].y = __expf((v[i].y - mx) * scale);
        v[i].z = __expf((v[i].z - mx) * scale); v[i].w = __expf((v[i].w - mx) * scale);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    _Float16* yr = out + (long)row * ld_out;
#pragma unroll
    for (int i = 0; i < SM_MAXQ; ++i) {
        const int q = tid + i * 256;
        if (q < Q) {
            f16x4 h = {(_Float16)(v[i].x * inv), (_Float16)(v[i].y * inv), (_Float16)(v[i].z * inv), (_Float16)(v[i].w * inv)};
            *reinterpret_cast<f16x4*>(yr + 4 * q) = h;
        }
    }
}

}  // namespace

extern "C" {

MLSD_API size_t mlsd_groupnorm_ws_bytes(int n_img, int HW, int n_grp)
{
    return (size_t)n_img * gn_chunks(HW, n_img) * n_grp * 2 * sizeof(float);
}

static int g_gn_finalize2 = 1;      /* two-level finalize on large maps (mlsd_groupnorm_set_finalize2: A/B timing, parity test) */
MLSD_API void mlsd_groupnorm_set_finalize2(int on) { g_gn_finalize2 = on ? 1 : 0; }

MLSD_API int mlsd_groupnorm(const mlsd_gn_args* a, void* stream)
{
    const int C = a->C1 + a->C2;
    if (a->n_grp <= 0 || a->n_grp > GN_MAXG || C % a->n_grp) return mlsd_set_error(-1, "mlsd_groupnorm: bad group count %d for C=%d", a->n_grp, C);
    const int cg = C / a->n_grp;
    if ((cg & 1) || (a->C1 & 3) || (a->C2 & 3)) return mlsd_set_error(-1, "mlsd_groupnorm: need even channels/group and C1,C2 %% 4 == 0 (C1=%d C2=%d cg=%d)", a->C1, a->C2, cg);
    if (C > 4 * 256 * GN_MAXQ) return mlsd_set_error(-1, "mlsd_groupnorm: C=%d too large", C);
    if ((a->ld1 & 3) || (a->C2 && (a->ld2 & 3))) return mlsd_set_error(-1, "mlsd_groupnorm: strides must be multiples of 4");
    if (!a->ws || !a->y16) return mlsd_set_error(-1, "mlsd_groupnorm: null workspace/output");
    GnP p;
    p.x1 = a->x1; p.x2 = a->x2; p.ld1 = a->ld1; p.ld2 = a->ld2; p.C1 = a->C1; p.C2 = a->C2; p.C = C;
    p.HW = a->HW; p.G = a->n_grp; p.cg = cg; p.eps = a->eps; p.gamma = a->gamma; p.beta = a->beta; p.silu = a->silu;
    p.y16 = (_Float16*)a->y16; p.raw16 = (_Float16*)a->raw16; p.ws = (float*)a->ws;
    p.nchunk = gn_chunks(a->HW, a->n_img);
    p.pix_per_chunk = (a->HW + p.nchunk - 1) / p.nchunk;
    p.nchunk = (a->HW + p.pix_per_chunk - 1) / p.pix_per_chunk;
    const dim3 grid(p.nchunk, a->n_img);
    p.cs1 = a->cs1; p.cs2 = a->cs2; p.rb1 = a->rb_rows1; p.rb2 = a->rb_rows2; p.mr = nullptr;
    const bool from_producers = a->cs1 && a->rb_rows1 > 0 && !(a->HW % a->rb_rows1) &&
                                (a->C2 == 0 || (a->cs2 && a->rb_rows2 > 0 && !(a->HW % a->rb_rows2)));
    int rc;
    if (gn_single_ok(a->n_img, a->HW, C, a->n_grp) && !from_producers && !((uintptr_t)a->gamma & 15) && !((uintptr_t)a->beta & 15)) {
        hipLaunchKernelGGL(gn_small_kernel, dim3(p.G, a->n_img), dim3(256), 0, (hipStream_t)stream, p);
        return mlsd_check_launch("gn_small_kernel");
    }
    if (from_producers) {
        // large maps: two-level finalize (coalesced); the chunk partials sit behind the mean / rstd table in the workspace (mlsd_groupnorm_ws_bytes: >= 64 chunks' worth)
        const int nrb_min = a->C2 ? (a->HW / a->rb_rows1 < a->HW / a->rb_rows2 ? a->HW / a->rb_rows1 : a->HW / a->rb_rows2) : a->HW / a->rb_rows1;
        const size_t mr_bytes = ((size_t)a->n_img * p.G * 2 * sizeof(float) + 15) & ~(size_t)15;
        const size_t wsb = mlsd_groupnorm_ws_bytes(a->n_img, a->HW, a->n_grp);
        int nch = wsb > mr_bytes ? (int)((wsb - mr_bytes) / ((size_t)a->n_img * p.G * 2 * sizeof(double))) : 0;
        if (nch > 64) nch = 64;
        if (g_gn_finalize2 && nrb_min >= 1024 && C <= GNF_MAXC && nch >= 16) {
            double* part = reinterpret_cast<double*>(reinterpret_cast<char*>(p.ws) + mr_bytes);
            hipLaunchKernelGGL(gn_finalize_l1, dim3(nch, a->n_img), dim3(256), 0, (hipStream_t)stream, p, part, nch);
            hipLaunchKernelGGL(gn_finalize_l2, dim3(a->n_img), dim3(256), 0, (hipStream_t)stream, p, (const double*)part, nch, p.ws);
        } else
        hipLaunchKernelGGL(gn_finalize, dim3(p.G, a->n_img), dim3(256), 0, (hipStream_t)stream, p, p.ws);
        rc = mlsd_check_launch("gn_finalize");
        p.mr = p.ws;
    } else {
        hipLaunchKernelGGL(gn_stats, grid, dim3(256), 0, (hipStream_t)stream, p);
        rc = mlsd_check_launch("gn_stats");
    }
    if (rc) return rc;
    hipLaunchKernelGGL(gn_apply, grid, dim3(256), (size_t)C * 2 * sizeof(float), (hipStream_t)stream, p);
    return mlsd_check_launch("gn_apply");
}

/* 1 if mlsd_groupnorm runs these dimensions as ONE dispatch (slab of an (image, group) held in registers): the plan builder then leaves the producers' column
 * statistics off for this GroupNorm (finalize + apply would be two dispatches) */
MLSD_API int mlsd_groupnorm_single_pass(int n_img, int HW, int C, int n_grp) { return gn_single_ok(n_img, HW, C, n_grp) ? 1 : 0; }
MLSD_API void mlsd_groupnorm_set_single(int on) { g_gn_single = on ? 1 : 0; }   /* diagnostics / A-B timing */

MLSD_API int mlsd_layernorm(const float* x, int64_t ldx, int rows, int d, float eps, const float* gamma,
                            const float* beta, void* y16, float* y32, void* stream)
{
    if ((d & 3) || d > 64 * 4 * LN_MAXQ || (ldx & 3)) return mlsd_set_error(-1, "mlsd_layernorm: unsupported d=%d ldx=%ld", d, (long)ldx);
    if (rows <= 0) return 0;
    const int nq = (d / 4 + 63) / 64;
    static const int env_blocks = [] { const char* e = getenv("MLSD_LN_STREAM_BLOCKS"); return e && *e ? atoi(e) : -1; }();   // A/B timing of whole plans
    if (env_blocks >= 0) g_ln_stream_blocks = env_blocks;
    if (g_ln_stream_blocks > 0 && rows > 4 * g_ln_stream_blocks && nq >= 2 && nq <= 5) {       // several rows per wave: the streaming form
        const dim3 grid(g_ln_stream_blocks), block(256);
#define MLSD_LN_STREAM(NQ) hipLaunchKernelGGL(ln_stream_kernel<NQ>, grid, block, 0, (hipStream_t)stream, x, (long)ldx, rows, d, eps, gamma, beta, (_Float16*)y16, y32)
        switch (nq) { case 2: MLSD_LN_STREAM(2); break; case 3: MLSD_LN_STREAM(3); break; case 4: MLSD_LN_STREAM(4); break; default: MLSD_LN_STREAM(5); break; }
#undef MLSD_LN_STREAM
        return mlsd_check_launch("ln_stream_kernel");
    }
    hipLaunchKernelGGL(ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, d, eps, gamma,
                       beta, (_Float16*)y16, y32);
    return mlsd_check_launch("ln_kernel");
}

MLSD_API void mlsd_layernorm_stream_blocks(int n) { g_ln_stream_blocks = n; }   /* diagnostics / A-B timing: 0 = one row per wave always */

MLSD_API int mlsd_softmax_rows(const float* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols,
                               float scale, void* stream)
{
    if (rows <= 0) return 0;
    if (!(cols & 3) && cols <= 4 * 256 * SM_MAXQ && !(ld_in & 3) && !(ld_out & 3) && !((uintptr_t)in & 15) && !((uintptr_t)out & 7)) {
        hipLaunchKernelGGL(softmax_rows_reg_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, (_Float16*)out,
                           (long)ld_out, rows, cols, scale);
        return mlsd_check_launch("softmax_rows_reg_kernel");
    }
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, (_Float16*)out,
                       (long)ld_out, rows, cols, scale);
    return mlsd_check_launch("softmax_rows_kernel");
}

}  // extern "C"
