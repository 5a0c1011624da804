// Hardware-map probes for gfx950 (MI355X).
//
// These tiny kernels pin down, with exact integer data, the lane<->element maps
// the production kernels (gemm_conv.hip, attention.hip) rely on:
//   * v_mfma_f32_32x32x16_f16 A/B/C fragment layout
//   * the "accumulator tile as the next MFMA's B operand" permuted-k recipe
//   * ds_read_b64_tr_b16 (transposed LDS read) block semantics
// They are exercised by tests/test_hw_probe.py (-m gpu).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "common.hpp"

// raw dump: every lane stores what it holds, python decodes.
__global__ void probe_mfma_raw(const _Float16* __restrict__ A,   // [32][16] row-major
                               const _Float16* __restrict__ B,   // [16][32] row-major
                               float* __restrict__ Craw)         // [64 lanes][16 regs]
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = A[r * 16 + 8 * h + j];     // A[row r][k = 8h + j]
        b[j] = B[(8 * h + j) * 32 + r];   // B[k = 8h + j][col r]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) Craw[l * 16 + i] = c[i];
}

// Y[32x32] = A2[32x32] * X[32x32], X = A1[32x16] * B1[16x32] kept in registers
// (X's accumulator registers re-used as the B operand, permuted k order).
__global__ void probe_mfma_chain(const _Float16* __restrict__ A1,  // [32][16]
                                 const _Float16* __restrict__ B1,  // [16][32]
                                 const _Float16* __restrict__ A2,  // [32][32]
                                 float* __restrict__ Yraw)         // [64][16]
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = A1[r * 16 + 8 * h + j];
        b[j] = B1[(8 * h + j) * 32 + r];
    }
    f32x16 x = {0};
    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x, 0, 0, 0);
    f32x16 y = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f16x8 xb, a2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xb[j] = (_Float16)x[8 * s + j];
            const int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);  // row of X
            a2[j] = A2[r * 32 + k];
        }
        y = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, xb, y, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Yraw[l * 16 + i] = y[i];
}

// ds_read_b64_tr_b16: tile [8 rows][32 cols] f16 in LDS, row stride `stride` halfs.
// lane L (group g=L>>4, idx=L&15, q=idx>>2, p=idx&3) supplies &T[4*(g>>1) + q][16*(g&1) + 4*p].
__global__ void probe_tr_read(const _Float16* __restrict__ T, int stride, float* __restrict__ out)  // out [64][4]
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 64) lds[i] = (i % 64 < stride && (i / 64) < 8) ? T[(i / 64) * stride + (i % 64)] : (_Float16)0;
    __syncthreads();
    const int L = threadIdx.x, g = L >> 4, idx = L & 15, q = idx >> 2, p = idx & 3;
    const _Float16* ptr = &lds[(4 * (g >> 1) + q) * 64 + 16 * (g & 1) + 4 * p];
    h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)ptr);
    for (int j = 0; j < 4; ++j) out[L * 4 + j] = (float)v[j];
}

__global__ void probe_copy_f32(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) dst[i] = src[i];
}


// ---- operand-fill probe (round 3): how many bytes per clock a CU can pull from L2 / MALL when it does nothing else.
// One 512-thread block per CU (128 KB of LDS requested so that two cannot share one); every thread issues 8 16-byte loads per
// trip (64 KB per block and trip, the size of a 256x256x64 GEMM stage), rows of 128 bytes at stride `row_stride` bytes like a GEMM operand
// panel; trip t of block b starts at row ((b * span_rows / nblocks) + t * 512) % span_rows of its XCD-shared panel, so `span_rows`
// sets the working set (small: L2 hits; large: MALL / HBM).  mode 0: global_load_lds_dwordx4 (LDS-DMA); 1: global_load_dwordx4
// into registers (consumed by an xor); 2: registers + ds_write_b128.
template <int MODE>
__global__ __launch_bounds__(512) void probe_fill_kernel(const unsigned char* __restrict__ src, long row_stride, int span_rows, int trips,
                                                         unsigned long long* __restrict__ clocks, unsigned* __restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = (int)((long)blockIdx.x * span_rows / gridDim.x);
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // wave-instruction = 8 rows x 128 bytes (lane l -> row l >> 3, chunk l & 7), like the GEMM staging
            int row = r0 + t * 512 + (i * 8 + wave) * 8 + (lane >> 3);
            row %= span_rows;
            const unsigned char* p = src + (long)row * row_stride + (lane & 7) * 16;
            if constexpr (MODE == 0) {
                unsigned char* dst = smem + ((t & 1) * 65536) + (i * 8 + wave) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                v[i] = *reinterpret_cast<const uint4*>(p);
            }
        }
        if constexpr (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the previous trip's 8 loads have landed; this trip's stay in flight
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<uint4*>(smem + ((t & 1) * 65536) + (i * 8 + wave) * 1024 + lane * 16) = v[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (MODE != 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
    if (MODE != 1 && smem[tid * 16] == 0x5a && smem[65536 + tid] == 0xa5) sink[1] = 1;
}

// Matrix-pipe RATE probe (round 4): what the part sustains when the matrix pipes are the only thing busy.  Every wave holds two A and two B fragments taken from `src`
// (the caller decides the data: random values or zeros -- the power a v_mfma draws depends on the bits it multiplies) and issues `iters` rounds of 8 independent
// v_mfma_f32_16x16x32_f16 on 8 accumulators: no memory traffic, no LDS, no barriers inside the loop.  clocks[block] = shader clocks of the loop; the caller times the launch.
__global__ __launch_bounds__(512) void probe_mfma_rate(const _Float16* __restrict__ src, int iters, unsigned long long* __restrict__ clocks, float* __restrict__ sink)
{
    const int tid = threadIdx.x;
    const f16x8* s8 = reinterpret_cast<const f16x8*>(src) + (size_t)(blockIdx.x & 63) * 2048 + tid * 4;
    const f16x8 a0 = s8[0], a1 = s8[1], b0 = s8[2], b1 = s8[3];
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((j & 1) ? a1 : a0, (j & 2) ? b1 : b0, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    f32x4 t = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += acc[j];
    if (t[0] + t[1] + t[2] + t[3] == 1.2345678e33f) sink[0] = t[0];       // (keeps the loop alive)
}

// Co-issue probe (round 6): does a wave's vector work run in the shadow of its own (and its SIMD neighbour's) MFMAs?  Per iteration 8 slices of {one v_mfma_f32_32x32x16_f16
// (8 passes = 32 matrix clocks) if MF, NV vector instructions of kind VK on registers no MFMA touches}; a scheduling barrier between slices keeps the emitted order.
// VK: 1 v_fma_f32, 2 v_exp_f32, 3 v_pk_fma_f32, 4 v_cvt_pk_f16_f32, 5 v_max3_f32, 6 v_pk_add_f32, 7 v_add_f32, 8 v_dot2_f32_f16, 9 v_pk_mul_f32, 10 v_pk_fma_f16, 11 v_exp_f16, 12 v_lshl_add_u64, 13 v_mad_u64_u32, 14 v_mul_lo_u32, 15 v_add_u32, 16 v_add_co_u32 + v_addc_co_u32, 17 v_add_f64.  clocks[block] = shader clocks of the loop (s_memtime domain is avoided: readcyclecounter).
template <bool MF, int VK, int NV>
__global__ __launch_bounds__(512) void probe_coissue(const _Float16* __restrict__ src, int iters, unsigned long long* __restrict__ clocks, float* __restrict__ sink)
{
    const int tid = threadIdx.x;
    const f16x8* s8 = reinterpret_cast<const f16x8*>(src) + (size_t)(blockIdx.x & 63) * 2048 + (tid & 511) * 4;
    const f16x8 a0 = s8[0], a1 = s8[1], b0 = s8[2], b1 = s8[3];
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float x[8];
    f32x2 y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = (float)a0[j] * 0.01f; y[j] = f32x2{(float)b0[j] * 0.01f, (float)b1[j] * 0.01f}; }
    const float c1 = 0.999f, c2 = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (MF) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16((j & 1) ? a1 : a0, (j & 2) ? b1 : b0, acc[j & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int r = (j * NV + k) & 7;
                if constexpr (VK == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c1), "v"(c2));
                else if constexpr (VK == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
                else if constexpr (VK == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[r]) : "v"(y[(r + 4) & 7]));
                else if constexpr (VK == 4) { unsigned h; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x[r]), "v"(x[(r + 1) & 7])); asm volatile("" :: "v"(h)); }
                else if constexpr (VK == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(x[(r + 3) & 7]), "v"(x[(r + 5) & 7]));
                else if constexpr (VK == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[r]) : "v"(y[(r + 4) & 7]));
                else if constexpr (VK == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[r]) : "v"(c2));
                else if constexpr (VK == 8) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[r]) : "v"(x[(r + 3) & 7]), "v"(x[(r + 5) & 7]));
                else if constexpr (VK == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[r]) : "v"(y[(r + 4) & 7]));
                else if constexpr (VK == 10) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x[r]) : "v"(x[(r + 3) & 7]), "v"(x[(r + 5) & 7]));
                else if constexpr (VK == 11) asm volatile("v_exp_f16 %0, %0" : "+v"(x[r]));
                else if constexpr (VK == 12) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(y[r]) : "v"(y[(r + 4) & 7]));
                else if constexpr (VK == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[r]) : "v"(x[(r + 3) & 7]), "v"(x[(r + 5) & 7]) : "vcc");
                else if constexpr (VK == 14) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[r]) : "v"(x[(r + 3) & 7]));
                else if constexpr (VK == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[r]) : "v"(x[(r + 3) & 7]));
                else if constexpr (VK == 16) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(x[r]), "+v"(x[(r + 1) & 7]) : "v"(x[(r + 3) & 7]), "v"(x[(r + 5) & 7]) : "vcc");
                else if constexpr (VK == 17) asm volatile("v_add_f64 %0, %0, %1" : "+v"(y[r]) : "v"(y[(r + 4) & 7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[j][e];
#pragma unroll
    for (int j = 0; j < 8; ++j) t += x[j] + y[j].x + y[j].y;
    if (t == 1.2345678e33f) sink[0] = t;
}

extern "C" {

/* Co-issue probe: nblocks x nthreads (256 = one wave per SIMD, 512 = two); mf = 1: 8 MFMAs of 32x32x16 per iteration; vk / nv: kind and number of vector instructions behind
 * each MFMA slot (nv in {0, 3, 6}); clocks[nblocks] = shader clocks of the loop. */
MLSD_API int mlsd_probe_coissue(const void* src, int iters, int nblocks, int nthreads, int mf, int vk, int nv, void* clocks, void* sink, void* stream)
{
    if (!src || iters < 1 || nblocks < 1 || (nthreads != 256 && nthreads != 512)) return mlsd_set_error(-1, "mlsd_probe_coissue: bad arguments");
    const dim3 g(nblocks), b(nthreads);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* s = (const _Float16*)src; unsigned long long* c = (unsigned long long*)clocks; float* k = (float*)sink;
#define CO_(MF, VK, NV) if (mf == MF && vk == VK && nv == NV) { hipLaunchKernelGGL((probe_coissue<MF != 0, VK, NV>), g, b, 0, st, s, iters, c, k); return mlsd_check_launch("probe_coissue"); }
#define COK_(VK) CO_(0, VK, 3) CO_(1, VK, 3) CO_(0, VK, 6) CO_(1, VK, 6)
    CO_(1, 0, 0) COK_(1) COK_(2) COK_(3) COK_(4) COK_(5) COK_(6) COK_(7) COK_(8) COK_(9) COK_(10) COK_(11) COK_(12) COK_(13) COK_(14) COK_(15) COK_(16) COK_(17)
#undef COK_
#undef CO_
    return mlsd_set_error(-1, "mlsd_probe_coissue: no such variant");
}

/* nblocks x 512 threads (8 waves: two per SIMD), `iters` x 8 MFMAs of 16x16x32 per wave; src: >= 64 x 2048 x 16 bytes of fp16 operands; clocks[nblocks].
 * FLOP of a launch = nblocks * 8 * iters * 8 * 16384 */
MLSD_API int mlsd_probe_mfma_rate(const void* src, int iters, int nblocks, void* clocks, void* sink, void* stream)
{
    if (!src || iters < 1 || nblocks < 1) return mlsd_set_error(-1, "mlsd_probe_mfma_rate: bad arguments");
    static const bool one = [] { const char* e = getenv("MLSD_PROBE_ONE_WAVE"); return e && *e && *e != '0'; }();      /* diagnostics: ONE wave per SIMD (256 threads) */
    hipLaunchKernelGGL(probe_mfma_rate, dim3(nblocks), dim3(one ? 256 : 512), 0, (hipStream_t)stream, (const _Float16*)src, iters, (unsigned long long*)clocks, (float*)sink);
    return mlsd_check_launch("probe_mfma_rate");
}

MLSD_API int mlsd_probe_mfma_raw(const void* A, const void* B, void* Craw, void* stream)
{
    hipLaunchKernelGGL(probe_mfma_raw, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const _Float16*)A, (const _Float16*)B, (float*)Craw);
    return mlsd_check_launch("probe_mfma_raw");
}

MLSD_API int mlsd_probe_mfma_chain(const void* A1, const void* B1, const void* A2, void* Yraw, void* stream)
{
    hipLaunchKernelGGL(probe_mfma_chain, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const _Float16*)A1, (const _Float16*)B1, (const _Float16*)A2, (float*)Yraw);
    return mlsd_check_launch("probe_mfma_chain");
}

MLSD_API int mlsd_probe_tr_read(const void* T, int stride, void* out, void* stream)
{
    hipLaunchKernelGGL(probe_tr_read, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const _Float16*)T, stride, (float*)out);
    return mlsd_check_launch("probe_tr_read");
}

MLSD_API int mlsd_probe_copy(const void* src, void* dst, size_t nbytes, void* stream)
{
    hipLaunchKernelGGL(probe_copy_f32, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, nbytes / 16);
    return mlsd_check_launch("probe_copy");
}

/* operand-fill probe: nblocks x 512 threads, `trips` trips of 64 KB per block; clocks[nblocks] = s_memtime clocks of each block's loop */
MLSD_API int mlsd_probe_fill(int mode, const void* src, long row_stride, int span_rows, int trips, int nblocks, void* clocks, void* sink, void* stream)
{
    if (mode < 0 || mode > 2 || span_rows < 512 || nblocks < 1) return mlsd_set_error(-1, "mlsd_probe_fill: bad arguments");
    const size_t lds = 131072;
    auto go = [&](auto k) -> int {
        MLSD_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3(nblocks), dim3(512), lds, (hipStream_t)stream, (const unsigned char*)src, row_stride, span_rows, trips,
                           (unsigned long long*)clocks, (unsigned*)sink);
        return mlsd_check_launch("probe_fill");
    };
    return mode == 0 ? go(probe_fill_kernel<0>) : mode == 1 ? go(probe_fill_kernel<1>) : go(probe_fill_kernel<2>);
}

}  // extern "C"
