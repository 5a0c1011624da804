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

extern "C" {

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

}  // extern "C"
