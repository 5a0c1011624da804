// Shared device-side helpers and the error convention of the kernel shim.
// Kernel shim entry points return 0 on success (like ggml_status /
// ggml_backend_graph_compute, reference src/mlblock.c:301-307) and <0 on error;
// the message is retrievable with mlsd_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define MLSD_API __attribute__((visibility("default")))

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

extern "C" {
int mlsd_set_error(int code, const char* fmt, ...);
int mlsd_check_launch(const char* what);
}

#define MLSD_HIP_TRY(expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError(); /* reported here: do not leave it pending for the next launch check */ \
            return mlsd_set_error(-(int)e_ - 1000, "%s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)

// ---- small device helpers -------------------------------------------------
// Activations on the fast hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): they sit in GEMM
// epilogues where a libm-grade tanhf (~30 VALU instructions) costs a quarter of a short-K main loop.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * fast_sigmoid(x); }
// tanh-approximation GELU (ggml_gelu): 0.5 x (1 + tanh(u)) == x * sigmoid(2u), u = sqrt(2/pi) x (1 + 0.044715 x^2).
// Round 6: the constants folded into the exp2 argument -- sigmoid(2u) = 1 / (1 + 2^(x (k1 + k2 x^2))), k1 = -2 sqrt(2/pi) log2(e), k2 = 0.044715 k1 -- : mul, fma, mul, v_exp_f32,
// add, v_rcp_f32, mul = 5 full-rate instructions + 2 transcendentals instead of 9 + 2 (the build runs with -ffp-contract=off: c * x * (1 + a * x * x), * 2, * log2(e) stayed six
// separate instructions).  The GEGLU epilogue of the 256 x 256 ping-pong tile does this 64 times per lane with the matrix pipe idle (~13 % of the SDXL feed-forward launches).
__device__ __forceinline__ float gelu_tanh_f(float x)
{
    const float k1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k2 = 0.044715f * k1;
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(k2, x * x, k1));      // 2^(-2u log2 e): inf for very negative x (-> 0), 0 for large x (-> x)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_quick_f(float x) { return x * fast_sigmoid(1.702f * x); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
