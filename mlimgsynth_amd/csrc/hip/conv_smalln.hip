// 3x3 convolution with a HANDFUL of output channels (Cout <= 16) over a full-resolution channels-last image: the KL-VAE decoder's conv_out (128 -> 3,
// reference src/vae.c:163-165 through mlb_nn_conv2d, src/mlblock_nn.c:31-55) and TAESD's last layer (64 -> 3, src/tae.c:88-89).
//
// On the implicit-GEMM tiles these launches are the worst of a decode: N = 3 on a 256 x 128 tile wastes 97 % of the MFMAs AND -- every K tile re-gathers its A rows
// through the im2col addressing -- ran at 682 GB/s (SDXL batch 4, 4 194 304 x 3 x 1152: 1.65 ms for 1.07 GB of activations; profiles/r5_sdxl_b4_vae_decode_shape_table.txt).
// The op is a STREAMING op: every input pixel is needed once (9 taps x Cout multiply-adds per channel), so the bound is HBM, and the design follows from that:
//
//   * no workgroup-level structure at all: every WAVE owns a 16-pixel-wide, TR-row-tall strip of one image and walks down it; its only state is a ring of NR input rows
//     (20 pixels x Cin halfs each: the strip + its halo, 16-byte aligned for the DMA) in ITS slice of the LDS.  No barrier, no shared tile, no inter-wave hand-off --
//     a wave's own counted `s_waitcnt vmcnt` orders its LDS-DMA before its fragment reads (MI355X_MICROARCH.md, two-waves item 7).
//   * input rows go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 4 pixels of 128 channels), NR - 3 rows ahead of the row being computed;
//     pixels outside the image (zero padding, ragged right edge) and the two unused halo slots read a 16-byte zero page.  The LDS image of a DMA instruction is lane-linear,
//     so the bank-conflict XOR swizzle is applied to the SOURCE chunk index and again on the fragment read (as in gemm_conv.hip).
//   * the arithmetic is MFMA after all -- v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand (16 "rows" = output channels, the ones >= Cout are zero) and 16 pixels as
//     the B operand: D[cout][pixel] leaves every lane < 16 with the Cout results of ONE pixel in consecutive registers = consecutive floats of the NHWC output.  36 MFMAs of 16
//     cycles per 16 pixels (Cin = 128) are ~0.1 ms for the whole SDXL batch: a third of the HBM time, hidden by the other waves of the CU.
//   * all 9 x Cin/32 weight fragments (144 registers at Cin = 128) stay in registers for the life of the wave; fp32 accumulation over (kh, kw, cin) in that order.
//
// Same results as the implicit-GEMM path up to fp32 summation order (both round nothing but the fp16 operands); tests/test_kernels_gpu.py::test_conv2d_small_n holds it
// against orc_conv2d at the fp32-output bound (2e-5).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

struct SNP {
    const _Float16* A; long lda;        // NHWC fp16 activations, lda = halfs between consecutive pixels
    const _Float16* Wt; long ldb;       // [N][3][3][Cin] fp16
    const float* bias;
    float* C32; long ldc32;
    int n_img, H, W, N, TR, tiles_x, tiles_y;
};

__device__ uint4 g_sn_zero[1];           // 16 zero bytes: source of every padded / unused 16-byte chunk (LDS-DMA has no bounds check and no zero fill)

template <int N>
__device__ __forceinline__ void sn_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt immediate range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CIN, int NR>
__global__ __launch_bounds__(256, 2) void conv_smalln_kernel(const SNP p)
{
    constexpr int PIXB = CIN * 2;                    // bytes per pixel
    constexpr int CH = PIXB / 16;                    // 16-byte chunks per pixel: 16 | 8
    constexpr int PPI = 1024 / PIXB;                 // pixels per DMA instruction: 4 | 8
    constexpr int NI = (20 + PPI - 1) / PPI;         // DMA instructions per input row: 5 | 3
    constexpr int PXR = NI * PPI;                    // pixel slots per row buffer: 20 | 24 (local pixel l = image column x0 - 2 + l; the taps read l = 1 .. 18)
    constexpr int ROWB = PXR * PIXB;                 // 5120 | 3072
    constexpr int KS = CIN / 32;                     // MFMA k-steps per tap
    constexpr int AHEAD = NR - 3;                    // input rows in flight beyond the three being read
    static_assert(CIN == 128 || CIN == 64, "pixel = 256 or 128 bytes");
    static_assert(NR >= 4 && NR <= 8, "ring depth");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned char* ring = smem + wave * (NR * ROWB);

    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int img = b / p.tiles_y;
    const int x0 = tx * 64 + wave * 16, y0 = ty * p.TR;
    if (x0 >= p.W) return;                           // (no barrier anywhere below: waves leave independently)
    const int y1 = min(y0 + p.TR, p.H);
    const _Float16* zsrc = reinterpret_cast<const _Float16*>(g_sn_zero);
    const _Float16* Aimg = p.A + (long)img * p.H * p.W * p.lda;

    // swizzle of a pixel's 16-byte chunks: the 16 pixels a fragment read touches (same logical chunk) must land on 16 different bank quads
    auto swz = [](int pxl) { return CIN == 128 ? (pxl & 15) : ((pxl >> 1) & 7); };

    // ---- LDS-DMA of one input row into ring slot `slot` (exactly NI instructions, always: the counted waits below rely on it)
    int d_pxl[NI], d_coff[NI];                       // per instruction: this lane's local pixel and the SOURCE chunk's offset in halfs
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int pxl = t * PPI + lane / CH, cs = lane % CH;
        d_pxl[t] = pxl; d_coff[t] = (cs ^ swz(pxl)) * 8;
    }
    const uintptr_t zaddr = (uintptr_t)zsrc;
    auto issue_row = [&](int yi, int slot) {
        const bool rowok = (unsigned)yi < (unsigned)p.H;
        const uintptr_t rowa = (uintptr_t)(Aimg + (long)yi * p.W * p.lda);
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int gx = x0 - 2 + d_pxl[t];
            const bool ok = rowok && (unsigned)gx < (unsigned)p.W && d_pxl[t] >= 1 && d_pxl[t] <= 18;
            // ONE instruction per piece whatever the lanes need: the address is blended arithmetically (as a branch the compiler emitted two DMA instructions per piece,
            // one per side, and the counted waits below count instructions)
            const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
            const uintptr_t addr = ((rowa + ((uintptr_t)((long)gx * p.lda + d_coff[t]) << 1)) & m) | (zaddr & ~m);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr,
                                             (__attribute__((address_space(3))) void*)(ring + slot * ROWB + t * 1024), 16, 0, 0);
        }
    };

    // ---- prologue: rows y0-1 .. y0+NR-3 into slots 0 .. NR-2, then the weights; one full wait covers both
#pragma unroll
    for (int r = 0; r < NR - 1; ++r) issue_row(y0 - 1 + r, r);
    const int wi = lane & 15, kc = lane >> 4;
    const int wrow = wi < p.N ? wi : p.N - 1;
    f16x8 wf[9][KS];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f16x8 w = *reinterpret_cast<const f16x8*>(p.Wt + (long)wrow * p.ldb + t * CIN + 32 * s + 8 * kc);
            if (wi >= p.N) w = f16x8{0, 0, 0, 0, 0, 0, 0, 0};      // (a select, not a branch: lanes of absent output channels read the last row and drop it)
            wf[t][s] = w;
        }
    float bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = (p.bias && 4 * kc + v < p.N) ? p.bias[4 * kc + v] : 0.0f;
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the compiler's scoreboard sees it (the weights never cost a wait inside the loop)

    // fragment read offsets inside a row buffer: tap column kw, k-step s  (pixel j of the strip reads local pixel j + kw + 1)
    const int j = lane & 15;
    int foff[3][KS];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int pxl = j + kw + 1;
            foff[kw][s] = pxl * PIXB + (((4 * s + kc) ^ swz(pxl)) << 4);
        }
    const bool col_ok = x0 + j < p.W;
    float* outp = p.C32 + ((long)img * p.H * p.W + (long)y0 * p.W + x0 + j) * p.ldc32 + 4 * kc;

    int slot0 = 0;                                   // ring slot of input row y - 1
    for (int y = y0; y < y1; ++y) {
        // row y + NR - 2 goes into the slot row y - 2 has just left (this wave's reads of it were consumed by the MFMAs of the previous iteration)
        int sl = slot0 + NR - 1; if (sl >= NR) sl -= NR;
        issue_row(y + NR - 2, sl);
        // rows <= y + 1 landed: at most the AHEAD youngest rows' instructions may be outstanding.  The output stores issued in between count in vmcnt too -- as YOUNGER
        // operations than row y + 1's, so this immediate is conservative for any number of them (it then also waits for the oldest of the rows in flight).
        sn_wait_vmcnt<AHEAD * NI>();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int rs = slot0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const unsigned char* rowb = ring + rs * ROWB;
            // the 3 x KS fragments of an input row are read as one batch (12 ds_read_b128 in flight), then consumed: left to itself the compiler waits for every pair
            f16x8 px[3][KS];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int s = 0; s < KS; ++s) px[kw][s] = *reinterpret_cast<const f16x8*>(rowb + foff[kw][s]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kh * 3 + kw][s], px[kw][s], acc, 0, 0, 0);
            if (++rs == NR) rs = 0;
        }
        if (col_ok) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (4 * kc + v < p.N) outp[v] = acc[v] + bv[v];
        }
        outp += (long)p.W * p.ldc32;
        if (++slot0 == NR) slot0 = 0;
    }
    // (the rows issued beyond y1 land in this wave's own LDS slice; the wave may end with them in flight: LDS is released only when every wave of the block has ended
    // and the hardware retires a wave's outstanding memory operations first)
    sn_wait_vmcnt<0>();
}

bool sn_on()
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("MLSD_CONV_SMALLN"); on = (e && *e == '0') ? 0 : 1; }      // A/B switch: MLSD_CONV_SMALLN=0 keeps the implicit-GEMM tile
    return on != 0;
}
int g_sn_nr = 0, g_sn_tr = 0;      // 0 = automatic (below); tools/conv_smalln_bench.py sets them

}  // namespace

/* != 0 if this launch is one the small-Cout streaming convolution takes: 3x3, stride 1, pad 1, Cout <= 16, Cin 64 or 128 (channels-last fp16 source), fp32 output, bias only */
extern "C" MLSD_API int mlsd_conv_smalln_eligible(const mlsd_gemm_args* a)
{
    if (!a || !sn_on() || !a->conv || a->KH != 3 || a->KW != 3 || a->stride != 1 || a->pad != 1 || a->upsample) return 0;
    if (a->N < 1 || a->N > 16 || (a->Cin != 128 && a->Cin != 64) || a->OH != a->H || a->OW != a->W || a->M < 16384) return 0;
    if (!a->C32 || a->C16 || a->resid || a->rowbias || a->bias_m || a->act != MLSD_ACT_NONE || a->colstats || a->ln_y16 || a->gn_y16) return 0;
    if ((a->lda & 7) || a->lda < a->Cin || ((uintptr_t)a->A & 15) || ((uintptr_t)a->W_ & 15) || (a->ldb & 7) || a->ldb != 9L * a->Cin || a->ldc32 < a->N) return 0;
    return 1;
}

/* diagnostics (tools/conv_smalln_bench.py): ring depth 4..6 and strip height of the next launches */
extern "C" MLSD_API void mlsd_conv_smalln_set(int ring_rows, int strip_rows)
{
    g_sn_nr = (ring_rows >= 4 && ring_rows <= 6) ? ring_rows : 0;
    g_sn_tr = (strip_rows >= 1 && strip_rows <= 4096) ? strip_rows : 0;
}

extern "C" int mlsd_conv_smalln(const mlsd_gemm_args* a, void* stream)
{
    if (!mlsd_conv_smalln_eligible(a)) return mlsd_set_error(-1, "mlsd_conv_smalln: launch not eligible for the small-Cout streaming convolution");
    SNP p;
    p.A = (const _Float16*)a->A; p.lda = a->lda; p.Wt = (const _Float16*)a->W_; p.ldb = a->ldb; p.bias = a->bias; p.C32 = a->C32; p.ldc32 = a->ldc32;
    p.n_img = a->n_img; p.H = a->H; p.W = a->W; p.N = a->N;
    p.tiles_x = (a->W + 63) / 64;
    /* strip height: every strip re-reads 2 halo rows and pays one pipeline fill, so the tallest strip that still leaves >= 512 blocks (two per CU) -- measured on SDXL b4
     * (4 x 1024 x 1024 x 128): 240 us at 128 rows (512 blocks), 246 at 64, 254 at 16; on one 512 x 512 image: 21 us at 16 rows (256 blocks), 31 at 32, 96 at 128
     * (profiles/r6_conv_smalln_bench.txt) */
    int tr = g_sn_tr;
    if (tr <= 0) { tr = 128; while (tr > 8 && (long)p.tiles_x * ((a->H + tr - 1) / tr) * a->n_img < 512) tr >>= 1; }
    p.TR = tr < a->H ? tr : a->H;
    p.tiles_y = (a->H + p.TR - 1) / p.TR;
    const long nblk = (long)p.tiles_x * p.tiles_y * p.n_img;
    if (nblk > 0x7fffffffL) return mlsd_set_error(-1, "mlsd_conv_smalln: grid too large");
    const int nr = g_sn_nr ? g_sn_nr : (a->Cin == 128 ? 4 : 5);      /* ring depth: 8 waves per CU one row ahead at 256-byte pixels, two rows ahead at 128-byte pixels (measured) */
    auto go = [&](auto kfn, size_t lds) -> int {
        static thread_local const void* attr_done[8]; static thread_local int n_done = 0;      // (one host call per kernel, not per launch)
        bool seen = false;
        for (int i = 0; i < n_done; ++i) seen |= attr_done[i] == (const void*)kfn;
        if (!seen) {
            MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (n_done < 8) attr_done[n_done++] = (const void*)kfn;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, p);
        return mlsd_check_launch("conv_smalln_kernel");
    };
    if (a->Cin == 128) {
        switch (nr) {
        case 4: return go(conv_smalln_kernel<128, 4>, 4 * 4 * 5120);
        case 6: return go(conv_smalln_kernel<128, 6>, 4 * 6 * 5120);
        default: return go(conv_smalln_kernel<128, 5>, 4 * 5 * 5120);
        }
    }
    switch (nr) {
    case 4: return go(conv_smalln_kernel<64, 4>, 4 * 4 * 3072);
    case 6: return go(conv_smalln_kernel<64, 6>, 4 * 6 * 3072);
    default: return go(conv_smalln_kernel<64, 5>, 4 * 5 * 3072);
    }
}
