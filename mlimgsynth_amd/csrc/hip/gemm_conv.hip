// Implicit-GEMM Conv2d / Linear on MFMA for gfx950 (MI355X).
//
//   C[M,N] = A[M,K] . W[N,K]^T      fp16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16)
//
// A is either a plain row-major matrix (Linear; reference src/mlblock_nn.c:16-28) or an
// on-the-fly im2col gather over a channels-last image (Conv2d 3x3/1x1, stride 1/2, zero
// padding, optional nearest-2x upsample of the source; reference src/mlblock_nn.c:31-55,
// :105-126).  K is ordered (kh, kw, cin) with cin fastest so a 16-byte chunk of A is 8
// consecutive channels of one input pixel: coalesced NHWC loads, no im2col buffer in HBM.
//
// Structure (one block = WAVES_M x WAVES_N wavefronts of 64 lanes):
//   * BK = 64; A and B tiles are staged global -> registers -> LDS (the gather needs per-lane
//     zero-fill, so register staging), two LDS buffers, ONE barrier per K-tile: the loads of
//     tile t+1 are issued before the MFMAs of tile t and written to the other buffer after them.
//   * LDS rows are 128 B (64 halfs); the 16-byte chunk c of row r lives at slot c ^ ((r>>1)&7):
//     a ds_read_b128 fragment read (lanes = 32 different rows, same logical chunk) is
//     bank-conflict free on the 64-bank LDS (cdna_hip_programming.md T2).
//   * each wave owns a (TM*32) x (TN*32) sub-tile: TM*TN accumulators of 16 fp32 registers.
//   * blockIdx is remapped so that the blocks of one XCD (blockIdx % 8) walk a contiguous
//     range of tiles: neighbouring tiles share the A panel in that XCD's L2 (T1).
//   * epilogue fuses bias, per-batch-row bias (time embedding), activation, GEGLU gating,
//     fp32 residual add, and writes fp32 and/or fp16.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

constexpr int BK = 64;

struct GemmP {
    const _Float16* A;
    const _Float16* B;
    long lda, ldb;
    int M, N, K;
    // conv geometry
    int H, W, Cin, OH, OW, KH, KW, stride, pad, ups;
    // epilogue
    const float* bias;
    const float* biasm;
    int act_post;
    const float* rowbias;
    int rows_per_batch;
    long ldrb;
    const float* resid;
    long ldr;
    int act;
    float* C32;
    long ldc32;
    _Float16* C16;
    long ldc16;
    int nbm, nbn;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(const GemmP p)
{
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM * 8 / THREADS, B_IT = BN * 8 / THREADS;
    constexpr int ROWS_PER_IT = THREADS / 8;
    static_assert(BM * 8 % THREADS == 0 && BN * 8 % THREADS == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                     // [2][BM][128 B]
    unsigned char* Bs = smem + 2 * BM * 128;      // [2][BN][128 B]

    // ---- XCD-aware tile mapping (bijective for any grid size)
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int bm = bid / p.nbn, bn = bid % p.nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- staging assignment: chunk column fixed per thread, A_IT/B_IT rows
    const int sc = tid & 7;        // 16-byte chunk (8 halfs) within the 64-wide K tile
    const int sr = tid >> 3;       // first row
    // conv gather state (per thread: position of its chunk in (kh,kw,cin); per row: pixel origin)
    int g_kh = 0, g_kw = 0, g_cin = sc * 8;
    int row_pix[A_IT], row_ih0[A_IT], row_iw0[A_IT];
    bool row_ok[A_IT];
    if (CONV) {
        while (g_cin >= p.Cin) { g_cin -= p.Cin; if (++g_kw == p.KW) { g_kw = 0; ++g_kh; } }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + sr + i * ROWS_PER_IT;
            row_ok[i] = m < p.M;
            const int mm = row_ok[i] ? m : 0;
            const int ohw = p.OH * p.OW;
            const int img = mm / ohw, rem = mm - img * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            row_pix[i] = img * p.H * p.W;
            row_ih0[i] = oh * p.stride - p.pad;
            row_iw0[i] = ow * p.stride - p.pad;
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = m0 + sr + i * ROWS_PER_IT;
            row_ok[i] = m < p.M;
            row_pix[i] = row_ok[i] ? m : 0;
            row_ih0[i] = row_iw0[i] = 0;
        }
    }
    const int He = p.ups ? p.H * 2 : p.H, We = p.ups ? p.W * 2 : p.W;

    uint4 ra[A_IT], rb[B_IT];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tile = [&](int kt) {
        const int k = kt * BK + sc * 8;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            uint4 v = zero4;
            if (CONV) {
                const int ih = row_ih0[i] + g_kh, iw = row_iw0[i] + g_kw;
                if (row_ok[i] && g_kh < p.KH && (unsigned)ih < (unsigned)He && (unsigned)iw < (unsigned)We) {
                    const int sh = p.ups ? (ih >> 1) : ih, sw = p.ups ? (iw >> 1) : iw;
                    const long off = (long)(row_pix[i] + sh * p.W + sw) * p.lda + g_cin;
                    v = *reinterpret_cast<const uint4*>(p.A + off);
                }
            } else {
                if (row_ok[i] && k < p.K) v = *reinterpret_cast<const uint4*>(p.A + (long)row_pix[i] * p.lda + k);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + sr + i * ROWS_PER_IT;
            uint4 v = zero4;
            if (n < p.N && k < p.K) v = *reinterpret_cast<const uint4*>(p.B + (long)n * p.ldb + k);
            rb[i] = v;
        }
        if (CONV) {  // advance this thread's chunk by one K tile
            g_cin += BK;
            while (g_cin >= p.Cin) { g_cin -= p.Cin; if (++g_kw == p.KW) { g_kw = 0; ++g_kh; } }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = sr + i * ROWS_PER_IT;
            *reinterpret_cast<uint4*>(As + buf * BM * 128 + lds_off(r, sc)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = sr + i * ROWS_PER_IT;
            *reinterpret_cast<uint4*>(Bs + buf * BN * 128 + lds_off(r, sc)) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        const unsigned char* Ab = As + cur * BM * 128;
        const unsigned char* Bb = Bs + cur * BN * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f16x8 af[TM], bf[TN];
            const int ch = ks * 2 + lh;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f16x8*>(Ab + lds_off(wm * WM + i * 32 + lr, ch));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb + lds_off(wn * WN + j * 32 + lr, ch));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][e]: row = (e&3) + 8*(e>>2) + 4*lh, col = lr  (probe-verified map)
    const bool geglu = p.act == MLSD_ACT_GEGLU;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (m >= p.M) continue;
            const float* rbias = p.rowbias ? p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb : nullptr;
            if (!geglu) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WN + j * 32 + lr;
                    if (n >= p.N) continue;
                    float v = acc[i][j][e];
                    if (p.bias) v += p.bias[n];
                    if (p.biasm) v += p.biasm[m];
                    if (rbias) v += rbias[n];
                    if (p.act_post && p.resid) v += p.resid[(long)m * p.ldr + n];
                    switch (p.act) {
                    case MLSD_ACT_SILU: v = silu_f(v); break;
                    case MLSD_ACT_GELU: v = gelu_tanh_f(v); break;
                    case MLSD_ACT_GELU_QUICK: v = gelu_quick_f(v); break;
                    case MLSD_ACT_RELU: v = fmaxf(v, 0.f); break;
                    default: break;
                    }
                    if (!p.act_post && p.resid) v += p.resid[(long)m * p.ldr + n];
                    if (p.C32) p.C32[(long)m * p.ldc32 + n] = v;
                    if (p.C16) p.C16[(long)m * p.ldc16 + n] = (_Float16)v;
                }
            } else {
                // weight rows interleaved in blocks of 32: tile j even = value, j odd = gate
                if constexpr (TN % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < TN; j += 2) {
                        const int nv = n0 + wn * WN + j * 32 + lr, ng = nv + 32;
                        if (ng >= p.N) continue;
                        float v = acc[i][j][e], g = acc[i][j + 1][e];
                        if (p.bias) { v += p.bias[nv]; g += p.bias[ng]; }
                        v = v * gelu_tanh_f(g);
                        const int no = ((n0 + wn * WN + j * 32) >> 6) * 32 + lr;
                        if (p.resid) v += p.resid[(long)m * p.ldr + no];
                        if (p.C32) p.C32[(long)m * p.ldc32 + no] = v;
                        if (p.C16) p.C16[(long)m * p.ldc16 + no] = (_Float16)v;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch(const mlsd_gemm_args* a, hipStream_t st)
{
    GemmP p;
    p.A = (const _Float16*)a->A; p.B = (const _Float16*)a->W_;
    p.lda = a->lda; p.ldb = a->ldb; p.M = a->M; p.N = a->N; p.K = a->K;
    p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.OH = a->OH; p.OW = a->OW; p.KH = a->KH; p.KW = a->KW;
    p.stride = a->stride; p.pad = a->pad; p.ups = a->upsample;
    p.bias = a->bias; p.biasm = a->bias_m; p.act_post = a->act_after_resid; p.rowbias = a->rowbias; p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
    p.ldrb = a->ldrb; p.resid = a->resid; p.ldr = a->ldr; p.act = a->act;
    p.C32 = a->C32; p.ldc32 = a->ldc32; p.C16 = (_Float16*)a->C16; p.ldc16 = a->ldc16;
    p.nbm = (a->M + BM - 1) / BM; p.nbn = (a->N + BN - 1) / BN;
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    constexpr size_t LDS = (size_t)2 * (BM + BN) * 128;
    const dim3 grid(p.nbm * p.nbn), block(THREADS);
    if (a->conv) {
        auto kfn = gemm_kernel<BM, BN, WAVES_M, WAVES_N, true>;
        if (LDS > 65536) MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        hipLaunchKernelGGL(kfn, grid, block, LDS, st, p);
    } else {
        auto kfn = gemm_kernel<BM, BN, WAVES_M, WAVES_N, false>;
        if (LDS > 65536) MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        hipLaunchKernelGGL(kfn, grid, block, LDS, st, p);
    }
    return mlsd_check_launch("gemm_kernel");
}

int pick_variant(const mlsd_gemm_args* a)
{
    // 0: 128x128 tile, 2x2 waves (wave tile 64x64)   1: 64x128 tile, 2x2 waves (wave tile 32x64, small M)
    if (a->M <= 64) return 1;
    return 0;
}

}  // namespace

extern "C" {

MLSD_API int mlsd_gemm(const mlsd_gemm_args* a, void* stream)
{
    if (!a || !a->A || !a->W_) return mlsd_set_error(-1, "mlsd_gemm: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return mlsd_set_error(-1, "mlsd_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    if ((a->K & 7) || (a->lda & 7) || (a->ldb & 7)) return mlsd_set_error(-1, "mlsd_gemm: K, lda, ldb must be multiples of 8 (K=%d lda=%ld ldb=%ld)", a->K, (long)a->lda, (long)a->ldb);
    if (((uintptr_t)a->A & 15) || ((uintptr_t)a->W_ & 15)) return mlsd_set_error(-1, "mlsd_gemm: operands must be 16-byte aligned");
    if (a->conv) {
        if (a->K != a->KH * a->KW * a->Cin) return mlsd_set_error(-1, "mlsd_gemm: conv K mismatch");
        if (a->Cin & 7) return mlsd_set_error(-1, "mlsd_gemm: conv Cin must be a multiple of 8");
        if (a->M != a->n_img * a->OH * a->OW) return mlsd_set_error(-1, "mlsd_gemm: conv M mismatch");
    }
    if (a->act == MLSD_ACT_GEGLU && (a->N & 63)) return mlsd_set_error(-1, "mlsd_gemm: GEGLU needs N %% 64 == 0");
    if (!a->C32 && !a->C16) return mlsd_set_error(-1, "mlsd_gemm: no output");
    hipStream_t st = (hipStream_t)stream;
    switch (pick_variant(a)) {
    case 1: return launch<64, 128, 2, 2>(a, st);   // small M: wave tile 32x64
    default: return launch<128, 128, 2, 2>(a, st);
    }
}

MLSD_API const char* mlsd_gemm_variant(const mlsd_gemm_args* a)
{
    switch (pick_variant(a)) {
    case 1: return a->conv ? "gemm_kernel<64,128,2,2,conv>" : "gemm_kernel<64,128,2,2,linear>";
    default: return a->conv ? "gemm_kernel<128,128,2,2,conv>" : "gemm_kernel<128,128,2,2,linear>";
    }
}

}  // extern "C"
