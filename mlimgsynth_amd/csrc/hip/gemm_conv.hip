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
// Two kernel families share this file's launcher, argument struct and tile-variant table:
//   gemm_pp_kernel (gemm_pp.hpp)  persistent two-group "ping-pong" tiles, register-direct epilogue: the default for
//                                 every problem made of whole wave blocks (linear, and conv with Cin % 64 == 0);
//   gemm_kernel (below)           general tiles: ragged M/N/K, padded channels, upsampled conv sources, split-K, GEGLU
//                                 on narrow outputs, unaligned layouts.
//
// Structure of gemm_kernel (one block = WAVES_M x WAVES_N wavefronts of 64 lanes):
//   * A and B tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no
//     ds_write); padded / out-of-range chunks read a 16-byte zero page.  The LDS image is lane-linear
//     per wave-instruction, so the bank-conflict XOR swizzle is applied to the SOURCE chunk index
//     (cdna_hip_programming.md rule 21) and again on the fragment read: ds_read_b128 of 32 rows
//     x one logical chunk touches every bank once.
//   * NSTAGE-deep LDS ring, ONE raw s_barrier per K tile, counted s_waitcnt vmcnt(N): the loads of the
//     next NSTAGE-2 tiles stay in flight across the barrier (never drained to 0 inside the loop).
//       iteration kt:  wait(tile kt landed) ; barrier ; issue tile kt+NSTAGE-1 into the slot that
//                      tile kt-1 just vacated ; MFMAs on tile kt
//     RAW: a wave's own counted wait + the barrier order every wave's DMA before any fragment read.
//     WAR: a slot is refilled only after the barrier that every wave reaches after its reads of it.
//   * each wave owns a (TM*32) x (TN*32) sub-tile: TM*TN accumulators of 16 fp32 registers.
//   * blockIdx is remapped so that the blocks of one XCD (blockIdx % 8) walk a contiguous
//     range of tiles: neighbouring tiles share the A panel in that XCD's L2 (T1).
//   * epilogue: each wave transposes its slab through LDS so a lane owns 4 consecutive columns;
//     bias, per-image row bias (time embedding), activation, GEGLU gating, fp32 residual and the
//     fp32 / fp16 stores are 16-byte / 8-byte accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>
#include "common.hpp"
#include "mlsd_kernels.h"

namespace {

struct GemmP {
    const _Float16* A;
    const _Float16* B;
    long lda, ldb;
    int M, N, K;
    // conv geometry
    int H, W, Cin, OH, OW, KH, KW, stride, pad, ups;
    int korder;   // conv, Cin % 64 == 0: 0 = K runs (kh, kw, cin), 1 = (cin / 64, kh, kw, cin % 64): the KH KW taps of a 64-channel slab are consecutive K tiles (mlsd_gemm_args.conv_korder)
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
    int vec;   // 1: every row stride and N are multiples of 4 -> wide (LDS-transposed) epilogue
    int dbg;   // diagnostics (timing-only builds of the loop): bit0 = no refills in the loop, bit1 = no MFMA/ds_read
    // split-K: blockIdx.y = slice z owns K tiles [z*kt_per, (z+1)*kt_per) and writes its raw fp32 partial sums to
    // C32 + z*ws_stride (the epilogue fields are cleared by the launcher; splitk_reduce applies them)
    int kt_per;
    long ws_stride;
    int gw;    // tile-order panel width (0: row-major)
    unsigned long long* tbuf;   // diagnostics: per-block cycle stamps of the ping-pong kernels (tools/gemm_trace.py), or null
    float* colstats;            // ping-pong kernels built with a *_STATS epilogue: [row block][2][N] column sums / sums of squares
    // stream-K (gemm_pp.hpp, SK): K-tile units per block, slabs [block][BM*BN] fp32, one flag per block
    int sk_L;
    float* sk_ws;
    unsigned* sk_flag;
    // LayerNorm at the end of the launch (gemm_pp.hpp *_LN epilogues): gamma, beta, eps, fp16 output, per-(row block, tile column, row) partials, counters
    const float *ln_g, *ln_b;
    float ln_eps;
    _Float16* ln_y;
    long ldln;
    float* ln_ws;
    unsigned* ln_cnt;
    int ln_slot;                // index of this launch's epoch word in ln_cnt (round 6: self-tagged records, no counters)
    int gn_G, gn_hw, gn_silu;   // splitk_reduce_gn: groups, rows per image, SiLU (gamma / beta / eps / output in the ln_* fields)
    // cross attention at the end of its q projection (gemm_pp.hpp PP_EPI_XATTN): K [n_img * Tk][ldk], V^T [n_img][N][96], output, rows per image, keys, log2(e) / sqrt(64)
    const _Float16 *xa_k, *xa_vt;
    _Float16* xa_out;
    long xa_ldk, xa_ldo;
    int xa_Tq, xa_Tk;
    float xa_sc;
};

// LDS tile: rows of BK halfs (128 B at BK=64, 64 B at BK=32); the 16-byte chunk c of row r lives at slot
// c ^ swz(r) so that a ds_read_b128 fragment read (32 lanes = 32 consecutive rows, one logical chunk)
// touches every bank once (64 banks x 4 B; 16-lane service groups).
template <int BK>
__device__ __forceinline__ int row_swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ row_swz<BK>(row)) << 4); }

// 16 zero bytes in global memory: the source of every padded / out-of-range 16-byte chunk
// (global_load_lds has no bounds check and no zero-fill)
__device__ uint4 g_zero_page[4];

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt immediate range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PAR (round 4): a split-K launch that REDUCES ITS SLICES ITSELF, in parallel.  SD1.5 batch 1 is bound by its dispatch count (~4.5 us per dependent dispatch, 498 per
// evaluation, 96 of them splitk_reduce).  Every block writes its raw partial tile to its slice of the workspace with write-through stores (as the two-launch form does),
// takes a ticket on the TILE's arrival counter, and one thread waits (bounded) until all K slices of the tile have arrived -- the whole grid is resident: the launcher
// checks tiles x slices against the occupancy of this kernel.  Then the blocks of the tile SHARE the reduction: block z adds, for its 1/nsl of the tile's 4-column
// groups, the slices in order 0, 1, ... (agent-scope loads) and applies the epilogue -- the same operations in the same order as splitk_reduce: bit-identical to the
// two-launch form.  (The round-3 variant let the LAST block reduce the whole tile alone at one CU's load bandwidth and lost to the second launch; profiles/NOTES.md.)
// A departure counter clears both words for the next launch; a give-up raises the sticky word sk_flag[4095] (mlctx_handoff_check -> retry on the two-launch plan).
// `pe` = the epilogue as requested (p carries the raw-partial form).
// ST: the build whose wide epilogue also emits column statistics (GemmP::colstats).  A build of its own: the 8 + 8 running sums push the 512- / 1024-thread tiles over
// 128 registers (120 -> 130: three waves per SIMD instead of four, the VAE's 256x128 convolutions +5 % -- measured with the sums in the common build); here the ST builds
// of those tiles are held to four waves per SIMD and spill a few registers in the epilogue instead.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool CONV, int NSTAGE, int DBG = 0, bool REG = false, bool PAR = false, bool ST = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) __attribute__((amdgpu_waves_per_eu((ST && WAVES_M * WAVES_N >= 8) ? 4 : 1, (ST && WAVES_M * WAVES_N >= 8) ? 4 : 10)))
void gemm_kernel(const GemmP p, const GemmP pe)
{
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int CPR = BK / 8;                   // 16-byte chunks per tile row
    constexpr int RB = BK * 2;                    // bytes per tile row
    constexpr int A_IT = BM * CPR / THREADS, B_IT = BN * CPR / THREADS;
    constexpr int LPT = A_IT + B_IT;              // LDS-DMA instructions per thread per K tile
    constexpr int ROWS_PER_IT = THREADS / CPR;
    constexpr int ROWS_PER_WAVE = 64 / CPR;       // rows one wave-instruction covers
    constexpr int STAGE_BYTES = (BM + BN) * RB;
    static_assert(BM * CPR % THREADS == 0 && BN * CPR % THREADS == 0, "tile/threads mismatch");
    static_assert(ROWS_PER_IT % 16 == 0, "row swizzle must be invariant over a thread's rows");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NSTAGE][A tile | B tile]; ONE LDS object

    // ---- XCD-aware tile mapping (bijective for any grid size)
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    // Tile order inside the XCD's contiguous range: column panels of p.gw tiles, rows fastest inside a panel,
    // so the ~32 blocks an XCD runs concurrently cover a (32/gw) x gw patch: per K step the XCD's L2 takes
    // 32/gw + gw distinct operand tiles instead of 1 + 32 (row-major order on a wide output).
    int bm, bn;
    if (p.gw > 0 && p.nbn > p.gw) {
        const int per_panel = p.gw * p.nbm;
        const int panel = bid / per_panel;
        const int first = panel * p.gw;
        const int w = min(p.gw, p.nbn - first);
        const int r = bid - panel * per_panel;
        bm = r / w; bn = first + (r - bm * w);
    } else {
        bm = bid / p.nbn; bn = bid - bm * p.nbn;
    }
    const int m0 = bm * BM, n0 = bn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- staging assignment.  A wave-instruction writes 1 KiB of LDS linearly (lane l -> row l/CPR, slot
    // l%CPR); the thread therefore fetches the LOGICAL chunk whose swizzled slot is tid%CPR.  row_swz is the
    // same for all of a thread's rows (they are ROWS_PER_IT, a multiple of 16, apart).
    const int sr = tid / CPR;
    const int sc = REG ? (tid % CPR) : ((tid % CPR) ^ row_swz<BK>(sr));   // REG: swizzle applied on the ds_write instead
    const int nkt_all = (p.K + BK - 1) / BK;
    const int kt0 = blockIdx.y * p.kt_per;        // split-K slice (kt_per == nkt_all, gridDim.y == 1 when not split)
    const int nkt = min(p.kt_per, nkt_all - kt0);
    int g_kh = 0, g_kw = 0, g_cin = kt0 * BK + sc * 8;   // conv: position of this thread's chunk in (kh, kw, cin)
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
    const _Float16* zsrc = reinterpret_cast<const _Float16*>(g_zero_page);

    // issue the LDS-DMA of K tile kt into ring slot `slot` (exactly LPT instructions per thread, always)
    auto stage_tile = [&](int kt, int slot) {
        const int k = (kt0 + kt) * BK + sc * 8;
        unsigned char* As = smem + slot * STAGE_BYTES;
        unsigned char* Bs = As + BM * RB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const _Float16* src = zsrc;
            if (CONV) {
                const int ih = row_ih0[i] + g_kh, iw = row_iw0[i] + g_kw;
                if (row_ok[i] && g_kh < p.KH && (unsigned)ih < (unsigned)He && (unsigned)iw < (unsigned)We) {
                    const int sh = p.ups ? (ih >> 1) : ih, sw = p.ups ? (iw >> 1) : iw;
                    src = p.A + (long)(row_pix[i] + sh * p.W + sw) * p.lda + g_cin;
                }
            } else {
                if (row_ok[i] && k < p.K) src = p.A + (long)row_pix[i] * p.lda + k;
            }
            unsigned char* dst = As + (wave * ROWS_PER_WAVE + i * ROWS_PER_IT) * RB;   // wave-uniform base
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + sr + i * ROWS_PER_IT;
            const _Float16* src = (n < p.N && k < p.K) ? p.B + (long)n * p.ldb + k : zsrc;
            unsigned char* dst = Bs + (wave * ROWS_PER_WAVE + i * ROWS_PER_IT) * RB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        if (CONV) {  // advance this thread's chunk by one K tile
            g_cin += BK;
            while (g_cin >= p.Cin) { g_cin -= p.Cin; if (++g_kw == p.KW) { g_kw = 0; ++g_kh; } }
        }
    };

    // register-staged alternative (REG): global_load_dwordx4 -> VGPR -> ds_write_b128.  The LDS-DMA path is
    // limited by its issue rate (~64 cycles per 1 KiB wave-instruction per CU, measured: the whole 256x256
    // kernel runs at exactly that rate); ordinary loads return at the L1 rate and the ds_write costs ~13 cycles.
    uint4 ra[A_IT], rb[B_IT];
    auto load_tile_reg = [&](int kt) {
        const int k = (kt0 + kt) * BK + sc * 8;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (CONV) {
                const int ih = row_ih0[i] + g_kh, iw = row_iw0[i] + g_kw;
                if (row_ok[i] && g_kh < p.KH && (unsigned)ih < (unsigned)He && (unsigned)iw < (unsigned)We) {
                    const int sh = p.ups ? (ih >> 1) : ih, sw = p.ups ? (iw >> 1) : iw;
                    v = *reinterpret_cast<const uint4*>(p.A + (long)(row_pix[i] + sh * p.W + sw) * p.lda + g_cin);
                }
            } else {
                if (row_ok[i] && k < p.K) v = *reinterpret_cast<const uint4*>(p.A + (long)row_pix[i] * p.lda + k);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + sr + i * ROWS_PER_IT;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n < p.N && k < p.K) v = *reinterpret_cast<const uint4*>(p.B + (long)n * p.ldb + k);
            rb[i] = v;
        }
        if (CONV) {
            g_cin += BK;
            while (g_cin >= p.Cin) { g_cin -= p.Cin; if (++g_kw == p.KW) { g_kw = 0; ++g_kh; } }
        }
    };
    auto store_tile_reg = [&](int slot) {
        unsigned char* As = smem + slot * STAGE_BYTES;
        unsigned char* Bs = As + BM * RB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(As + lds_off<BK>(sr + i * ROWS_PER_IT, sc)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<uint4*>(Bs + lds_off<BK>(sr + i * ROWS_PER_IT, sc)) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if constexpr (REG) {
        load_tile_reg(0);
        store_tile_reg(0);
    } else {
        // prologue: NSTAGE-1 tiles in flight
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nkt) stage_tile(s, s);
    }

    // Residual prefetch (tiles with register headroom: 128x320, 8 waves at <= 256 VGPRs): the fp32 residual tile is
    // as many bytes as the tile's output and, loaded in the epilogue, is pure exposed latency for a short-K GEMM
    // (every CU reaches its epilogue at the same time).  Issued here, right after the first K tile's staging loads,
    // it lands under the main loop.  Layout = the wide epilogue's (lane owns 4 consecutive columns); addresses are
    // clamped instead of predicated so that every wave issues exactly NPRE loads (the counted vmcnt below).
    constexpr bool PRE = (BM == 128 && BN == 320) && !REG;
    constexpr int NPRE = PRE ? TM * ((TN / 2) * 8 + (TN % 2) * 4) : 1;
    float4 rpre[NPRE];
    bool pre = false;
    if constexpr (PRE) {
        pre = p.resid && p.vec && p.act != MLSD_ACT_GEGLU;
        if (pre) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int jc = 0; jc < TN; jc += 2) {
                    const bool pair = jc + 1 < TN;
                    const int slab0 = i * ((TN / 2) * 8 + (TN % 2) * 4) + (jc / 2) * 8;
                    const int c4 = pair ? (lane & 15) * 4 : (lane & 7) * 4;
                    const int n = min(n0 + wn * WN + jc * 32 + c4, p.N - 4);
#pragma unroll
                    for (int it = 0; it < (pair ? 8 : 4); ++it) {
                        const int row = pair ? it * 4 + (lane >> 4) : it * 8 + (lane >> 3);
                        const int m = min(m0 + wm * WM + i * 32 + row, p.M - 1);
                        rpre[slab0 + it] = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
                    }
                }
            }
        }
    }

    int slot = 0;                                  // ring slot of tile kt
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (REG) {
            __syncthreads();                       // tile kt visible (its ds_writes precede this barrier)
            if (kt + 1 < nkt && !(DBG & 1)) load_tile_reg(kt + 1);   // in flight during the MFMAs of tile kt
        } else {
            // tiles kt+1 .. kt+NSTAGE-2 may stay in flight; tile kt must have landed
            const int ahead = min(NSTAGE - 2, nkt - 1 - kt);
            if (PRE && NSTAGE == 2 && pre && kt == 0) wait_vmcnt<NPRE>();   // tile 0 landed; the residual prefetch may still fly
            else if (NSTAGE >= 4 && ahead >= 2) wait_vmcnt<2 * LPT>();
            else if (NSTAGE >= 3 && ahead == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // raw barrier: no implicit vmcnt(0) drain
            if (kt + NSTAGE - 1 < nkt && !(DBG & 1)) {
                int fill = slot + NSTAGE - 1; if (fill >= NSTAGE) fill -= NSTAGE;
                stage_tile(kt + NSTAGE - 1, fill); // refill the slot tile kt-1 vacated (all waves passed the barrier)
            }
        }
        const unsigned char* Ab = smem + slot * STAGE_BYTES;
        const unsigned char* Bb = Ab + BM * RB;
        // fragments are double-buffered in registers: the ds_read_b128 of k-step ks+1 are issued before the
        // MFMAs of k-step ks, so LDS latency hides under matrix work instead of stalling every k-step
        constexpr int KS = BK / 16;                // k-steps of the 32x32x16 MFMA per tile
        f16x8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int ks, int b) {
            const int ch = ks * 2 + lh;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[b][i] = *reinterpret_cast<const f16x8*>(Ab + lds_off<BK>(wm * WM + i * 32 + lr, ch));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[b][j] = *reinterpret_cast<const f16x8*>(Bb + lds_off<BK>(wn * WN + j * 32 + lr, ch));
        };
        if constexpr (!(DBG & 2)) {
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) load_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
        }
        }
        if constexpr (REG) { if (kt + 1 < nkt && !(DBG & 1)) store_tile_reg(slot ^ 1); }   // other slot: last read in iteration kt-1
        if (++slot == NSTAGE) slot = 0;
    }
    __syncthreads();   // every wave is done with the ring: it becomes the epilogue's staging space

    // ---- split-K reduced IN the launch (p.sk_L = number of K slices, gridDim.y): every block writes its partial accumulators to
    // its slab (register order: 1 KiB per wave-instruction, write-through), then takes a ticket on the tile's counter; the block
    // that draws the last ticket adds ALL slabs of the tile in slice order 0, 1, ... (its own included, re-read: the sum does not
    // depend on which block came last -> bit-reproducible and equal to the two-launch form) and runs the epilogue.  Saves the
    // second launch (~6 us on an SD1.5-sized problem, where it is a third of the GEMM) and its round trip through HBM.
#ifdef MLSD_GEMM_EXPERIMENTS
    if (p.sk_L > 1) {
        const int nsl = p.sk_L;
        constexpr int SLAB4 = BM * BN / 4;                                  // float4 per slab
        f32x4* const base = reinterpret_cast<f32x4*>(p.sk_ws) + (long)blockIdx.x * nsl * SLAB4 + wave * (TM * TN * 4) * 64 + lane;
        {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)blockIdx.y * SLAB4 - lane), 0, (TM * TN * 4) * 64 * 16, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (((i * TN + j) * 4 + q) * 64 + lane) * 16, 0, 16);   // aux 16 = sc1: write-through
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const tick = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.sk_flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(nsl - 1);
            if (last) __hip_atomic_store(p.sk_flag + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            *tick = last;
        }
        __syncthreads();
        const int last = *tick;
        __syncthreads();                                                    // (the epilogue's staging space starts at smem[0])
        if (!last) return;
        // (the slabs are read with agent-scope loads instead of an acquire fence: the fence alone costs microseconds, profiles/r3_ln_fold.txt)
        const auto rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(base - lane), 0, nsl * SLAB4 * 16, 0x00020000);
        // NB slabs in flight at a time (a slab = TM TN 4 wave-loads of 1 KiB; one at a time is a chain of exposed memory latencies)
        constexpr int NR = TM * TN * 4, NB = NR <= 8 ? 4 : 2;
        for (int z0 = 0; z0 < nsl; z0 += NB) {
            f32x4 t[NB][NR];
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (z0 + b < nsl) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        t[b][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, (((z0 + b) * SLAB4 + r * 64 + lane) * 16), 0, 16));   // aux 16 = sc1
                }
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (z0 + b < nsl) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = acc[r / (TN * 4)][(r / 4) % TN][4 * (r & 3) + e];
                            acc[r / (TN * 4)][(r / 4) % TN][4 * (r & 3) + e] = (z0 + b == 0) ? t[b][r][e] : d + t[b][r][e];   // slice 0 starts the sum (as the two-launch form does)
                        }
                }
        }
    }
#endif  // MLSD_GEMM_EXPERIMENTS (split-K reduced in the launch)

    // ---- epilogue.  acc[i][j][e]: row = (e&3) + 8*(e>>2) + 4*lh, col = lr  (probe-verified map)
    const bool geglu = p.act == MLSD_ACT_GEGLU;
    float* const C32 = p.C32 ? p.C32 + (long)blockIdx.y * p.ws_stride : nullptr;
    if (p.vec) {
        // Wide epilogue: each wave transposes its accumulators, 32 rows x 64 (or a last 32) columns at a time,
        // through a private 8 KiB LDS region (the tile buffers are free after the main loop's last barrier) so
        // that a lane owns 4 CONSECUTIVE columns: bias/residual loads and the stores are 16-byte (fp32) / 8-byte
        // (fp16) accesses, 16 lanes cover 256 contiguous bytes of a row; 16 store instructions per 32x64 slab
        // instead of 64 scalar ones.
        float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
        // bias, row bias, residual, activation and both stores for 4 consecutive columns of row m
        auto finish4 = [&](int m, int n, float4 v, const float4 bv, const float4 rs_pre, const bool have_pre) -> float4 {
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.biasm) { const float b = p.biasm[m]; v.x += b; v.y += b; v.z += b; v.w += b; }
            if (p.rowbias) {
                const float4 r = *reinterpret_cast<const float4*>(p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb + n);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            float4 rs = make_float4(0, 0, 0, 0);
            if (have_pre) rs = rs_pre;
            else if (p.resid) rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
            if (p.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
            switch (p.act) {
            case MLSD_ACT_SILU: v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); break;
            case MLSD_ACT_GELU: v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w); break;
            case MLSD_ACT_GELU_QUICK: v.x = gelu_quick_f(v.x); v.y = gelu_quick_f(v.y); v.z = gelu_quick_f(v.z); v.w = gelu_quick_f(v.w); break;
            case MLSD_ACT_RELU: v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); break;
            default: break;
            }
            if (!p.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
            if constexpr (PAR) {      // raw partial sums, written through to the agent-coherent level (other XCDs' blocks read them in this launch)
                const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)C32, 0, 0x7fffffff, 0x00020000);
                const f32x4 w = {v.x, v.y, v.z, v.w};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rsw, (int)(((long)m * p.ldc32 + n) * 4), 0, 16);   // aux 16 = sc1
            } else if (C32) *reinterpret_cast<float4*>(C32 + (long)m * p.ldc32 + n) = v;
            if (p.C16) {
                f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                *reinterpret_cast<f16x4*>(p.C16 + (long)m * p.ldc16 + n) = h;
            }
            return v;
        };
        // column statistics for a consuming GroupNorm (GemmP::colstats, round 4 on these tiles): per 64-column slab, the lane's 4 columns summed over the wave's WM rows
        // (8 rows per 32-row block in the lane, then the 4 row groups of the wave by shuffles): one partial per (block of WM rows, column), fixed order
        float4 st_s[ST ? (TN + 1) / 2 : 1], st_q[ST ? (TN + 1) / 2 : 1];
        if constexpr (ST) {
#pragma unroll
            for (int q = 0; q < (TN + 1) / 2; ++q) { st_s[q] = make_float4(0, 0, 0, 0); st_q[q] = make_float4(0, 0, 0, 0); }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int jc = 0; jc < TN; jc += 2) {
                const bool pair = jc + 1 < TN;             // 64-column slab, or the last 32 columns of an odd TN
                const int cb = n0 + wn * WN + jc * 32;     // first column of the slab
                const int SLAB0 = PRE ? i * ((TN / 2) * 8 + (TN % 2) * 4) + (jc / 2) * 8 : 0;   // this slab's first entry in rpre[]
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * 64 + lr] = acc[i][jc][e];
                    if (pair) stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * 64 + 32 + lr] = acc[i][pair ? jc + 1 : jc][e];
                }
                __builtin_amdgcn_wave_barrier();
                if (!geglu && pair) {
                    const int c4 = (lane & 15) * 4;
                    const int n = cb + c4;
                    float4 bv = make_float4(0, 0, 0, 0);
                    if (p.bias && n < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = it * 4 + (lane >> 4);
                        const int m = m0 + wm * WM + i * 32 + row;
                        const float4 v = *reinterpret_cast<const float4*>(stg + row * 64 + c4);
                        float4 rp = make_float4(0, 0, 0, 0);
                        if constexpr (PRE) rp = rpre[SLAB0 + it];
                        if (m >= p.M || n >= p.N) continue;
                        const float4 o = finish4(m, n, v, bv, rp, PRE && pre);
                        if constexpr (ST) {
                            st_s[jc / 2].x += o.x; st_s[jc / 2].y += o.y; st_s[jc / 2].z += o.z; st_s[jc / 2].w += o.w;
                            st_q[jc / 2].x += o.x * o.x; st_q[jc / 2].y += o.y * o.y; st_q[jc / 2].z += o.z * o.z; st_q[jc / 2].w += o.w * o.w;
                        }
                    }
                } else if (!geglu) {
                    const int c4 = (lane & 7) * 4;
                    const int n = cb + c4;
                    float4 bv = make_float4(0, 0, 0, 0);
                    if (p.bias && n < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + (lane >> 3);
                        const int m = m0 + wm * WM + i * 32 + row;
                        const float4 v = *reinterpret_cast<const float4*>(stg + row * 64 + c4);
                        float4 rp = make_float4(0, 0, 0, 0);
                        if constexpr (PRE) rp = rpre[SLAB0 + it];
                        if (m >= p.M || n >= p.N) continue;
                        finish4(m, n, v, bv, rp, PRE && pre);
                    }
                } else if (pair) {
                    // slab columns 0..31 = value, 32..63 = gate (weight rows interleaved in blocks of 32)
                    const int c4 = (lane & 7) * 4;
                    const int nv = cb + c4, ng = nv + 32;
                    const int no = (cb >> 6) * 32 + c4;
                    float4 bvv = make_float4(0, 0, 0, 0), bgg = bvv;
                    if (p.bias && ng < p.N) { bvv = *reinterpret_cast<const float4*>(p.bias + nv); bgg = *reinterpret_cast<const float4*>(p.bias + ng); }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + (lane >> 3);
                        const int m = m0 + wm * WM + i * 32 + row;
                        const float4 a4 = *reinterpret_cast<const float4*>(stg + row * 64 + c4);
                        const float4 g4 = *reinterpret_cast<const float4*>(stg + row * 64 + 32 + c4);
                        if (m >= p.M || ng >= p.N) continue;
                        float4 v;
                        v.x = (a4.x + bvv.x) * gelu_tanh_f(g4.x + bgg.x);
                        v.y = (a4.y + bvv.y) * gelu_tanh_f(g4.y + bgg.y);
                        v.z = (a4.z + bvv.z) * gelu_tanh_f(g4.z + bgg.z);
                        v.w = (a4.w + bvv.w) * gelu_tanh_f(g4.w + bgg.w);
                        if (p.resid) {
                            const float4 rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + no);
                            v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                        }
                        if (C32) *reinterpret_cast<float4*>(C32 + (long)m * p.ldc32 + no) = v;
                        if (p.C16) {
                            f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                            *reinterpret_cast<f16x4*>(p.C16 + (long)m * p.ldc16 + no) = h;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (ST) {     // (launcher: fp32 output, even TN, no GEGLU, single K slice: mlsd_gemm_colstats_rows)
            const int rbk = (m0 + wm * WM) / WM;
            if (m0 + wm * WM < p.M) {
#pragma unroll
                for (int q = 0; q < TN / 2; ++q) {
                    float4 a = st_s[q], b = st_q[q];
#pragma unroll
                    for (int o = 16; o <= 32; o <<= 1) {
                        a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
                        b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
                    }
                    const int n = n0 + wn * WN + q * 64 + (lane & 15) * 4;
                    if (lane < 16 && n < p.N) {
                        float* st = p.colstats + (long)rbk * 2 * p.N + n;
                        *reinterpret_cast<float4*>(st) = a;
                        *reinterpret_cast<float4*>(st + p.N) = b;
                    }
                }
            }
        }
        if constexpr (PAR) {
            // ---- the slices of this tile are added by the blocks that computed them
            const int nsl = gridDim.y, z = blockIdx.y;
            unsigned* const arr = p.sk_flag + blockIdx.x;             // arrivals of this tile (2047 tiles at most: the launcher checks)
            unsigned* const dep = p.sk_flag + 2048 + blockIdx.x;      // departures
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this block's partial tile is at the coherent level
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(arr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nsl && ++spins < (1u << 21)) __builtin_amdgcn_s_sleep(2);
                if (spins >= (1u << 21)) __hip_atomic_store(p.sk_flag + 4095, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // reported, not hung (mlctx_handoff_check)
            }
            __syncthreads();
            const int rows = min(BM, p.M - m0), c4 = min(BN, p.N - n0) >> 2;      // (N % 4 == 0 on this path)
            const int items = rows * c4, per = (items + nsl - 1) / nsl;
            const int i1 = min(items, (z + 1) * per);
            const auto rsl = __builtin_amdgcn_make_buffer_rsrc((void*)p.C32, 0, 0x7fffffff, 0x00020000);   // slice 0; slice zz at + zz * ws_stride floats
            for (int it = z * per + tid; it < i1; it += THREADS) {
                const int r = it / c4, m = m0 + r, n = n0 + ((it - r * c4) << 2);
                const int off = (int)(((long)m * p.N + n) * 4);
                float4 v = make_float4(0, 0, 0, 0);
                for (int z0 = 0; z0 < nsl; z0 += 4) {                // 4 slices in flight, added in slice order (the order of splitk_reduce)
                    f32x4 t[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        t[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, off, (int)((long)min(z0 + b, nsl - 1) * p.ws_stride * 4), 16));
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (z0 + b < nsl) {
                            if (z0 + b == 0) v = make_float4(t[b][0], t[b][1], t[b][2], t[b][3]);
                            else { v.x += t[b][0]; v.y += t[b][1]; v.z += t[b][2]; v.w += t[b][3]; }
                        }
                }
                // the requested epilogue: the operations of splitk_reduce, in its order
                if (pe.bias) { const float4 b = *reinterpret_cast<const float4*>(pe.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (pe.biasm) { const float b = pe.biasm[m]; v.x += b; v.y += b; v.z += b; v.w += b; }
                if (pe.rowbias) {
                    const float4 rb = *reinterpret_cast<const float4*>(pe.rowbias + (long)(m / pe.rows_per_batch) * pe.ldrb + n);
                    v.x += rb.x; v.y += rb.y; v.z += rb.z; v.w += rb.w;
                }
                float4 rs = make_float4(0, 0, 0, 0);
                if (pe.resid) rs = *reinterpret_cast<const float4*>(pe.resid + (long)m * pe.ldr + n);
                if (pe.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
                switch (pe.act) {
                case MLSD_ACT_SILU: v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); break;
                case MLSD_ACT_GELU: v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w); break;
                case MLSD_ACT_GELU_QUICK: v.x = gelu_quick_f(v.x); v.y = gelu_quick_f(v.y); v.z = gelu_quick_f(v.z); v.w = gelu_quick_f(v.w); break;
                case MLSD_ACT_RELU: v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); break;
                default: break;
                }
                if (!pe.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
                if (pe.C32) *reinterpret_cast<float4*>(pe.C32 + (long)m * pe.ldc32 + n) = v;
                if (pe.C16) {
                    f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                    *reinterpret_cast<f16x4*>(pe.C16 + (long)m * pe.ldc16 + n) = h;
                }
            }
            // departures: the last block to leave the tile clears both counters for the next launch (nobody can still be polling: every block has passed the wait)
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(dep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)(nsl - 1)) {
                    __hip_atomic_store(arr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dep, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        return;
    }
    // scalar fallback (N or a stride not a multiple of 4: 3-channel image outputs, padded 4-channel latents)
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
                    if (C32) C32[(long)m * p.ldc32 + n] = v;
                    if (p.C16) p.C16[(long)m * p.ldc16 + n] = (_Float16)v;
                }
            } else {
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
                        if (C32) C32[(long)m * p.ldc32 + no] = v;
                        if (p.C16) p.C16[(long)m * p.ldc16 + no] = (_Float16)v;
                    }
                }
            }
        }
    }
}

#include "gemm_pp.hpp"
#include "gemm_skinny.hpp"

// ---- split-K second pass: sum the slices in fixed order (deterministic), then the same epilogue as above.
// One thread per 4 consecutive columns; only launched when the wide-epilogue alignment conditions hold.
__global__ __launch_bounds__(256) void splitk_reduce(const GemmP p, const float* __restrict__ ws, int nsplit)
{
    const int n4 = p.N >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
    float4 v = *reinterpret_cast<const float4*>(ws + (long)m * p.N + n);
    for (int z = 1; z < nsplit; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(ws + z * p.ws_stride + (long)m * p.N + n);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (p.biasm) { const float b = p.biasm[m]; v.x += b; v.y += b; v.z += b; v.w += b; }
    if (p.rowbias) {
        const float4 r = *reinterpret_cast<const float4*>(p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    float4 rs = make_float4(0, 0, 0, 0);
    if (p.resid) rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
    if (p.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
    switch (p.act) {
    case MLSD_ACT_SILU: v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); break;
    case MLSD_ACT_GELU: v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w); break;
    case MLSD_ACT_GELU_QUICK: v.x = gelu_quick_f(v.x); v.y = gelu_quick_f(v.y); v.z = gelu_quick_f(v.z); v.w = gelu_quick_f(v.w); break;
    case MLSD_ACT_RELU: v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); break;
    default: break;
    }
    if (!p.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
    if (p.C32) *reinterpret_cast<float4*>(p.C32 + (long)m * p.ldc32 + n) = v;
    if (p.C16) {
        f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<f16x4*>(p.C16 + (long)m * p.ldc16 + n) = h;
    }
}

// ---- split-K second pass that ALSO emits the column statistics of the rows it finishes (GemmP::colstats: [block of 32 rows][2][N] sums / sums of squares, for a
// consuming GroupNorm: its first pass over the fp32 map disappears).  A block = 32 rows x 256 columns: thread (row lane rl = t / 64, column group t % 64) finishes
// rows rl, rl + 4, ... with the operations of splitk_reduce in its order (bit-identical output, the slices' loads of the 8 rows in flight together), sums its 4
// columns over its 8 rows, and the 4 row lanes are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void splitk_reduce_stats(const GemmP p, const float* __restrict__ ws, int nsplit)
{
    __shared__ float4 sh_s[4][64], sh_q[4][64];
    const int t = threadIdx.x, cgi = t & 63, rl = t >> 6;
    const int n = (blockIdx.x * 64 + cgi) * 4, mb = blockIdx.y * 32;
    const bool colok = n < p.N;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int m = mb + rl + 4 * k;
        v[k] = (colok && m < p.M) ? *reinterpret_cast<const float4*>(ws + (long)m * p.N + n) : make_float4(0, 0, 0, 0);
    }
    for (int z = 1; z < nsplit; ++z) {
        float4 u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = mb + rl + 4 * k;
            u[k] = (colok && m < p.M) ? *reinterpret_cast<const float4*>(ws + z * p.ws_stride + (long)m * p.N + n) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k].x += u[k].x; v[k].y += u[k].y; v[k].z += u[k].z; v[k].w += u[k].w; }
    }
    float4 cs = make_float4(0, 0, 0, 0), cq = cs;
    float4 bv = make_float4(0, 0, 0, 0);
    if (p.bias && colok) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int m = mb + rl + 4 * k;
        if (!colok || m >= p.M) continue;
        float4 x = v[k];
        if (p.bias) { x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w; }
        if (p.biasm) { const float b = p.biasm[m]; x.x += b; x.y += b; x.z += b; x.w += b; }
        if (p.rowbias) {
            const float4 r = *reinterpret_cast<const float4*>(p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb + n);
            x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
        }
        float4 rs = make_float4(0, 0, 0, 0);
        if (p.resid) rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
        if (p.act_post) { x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w; }
        switch (p.act) {
        case MLSD_ACT_SILU: x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); break;
        case MLSD_ACT_GELU: x.x = gelu_tanh_f(x.x); x.y = gelu_tanh_f(x.y); x.z = gelu_tanh_f(x.z); x.w = gelu_tanh_f(x.w); break;
        case MLSD_ACT_GELU_QUICK: x.x = gelu_quick_f(x.x); x.y = gelu_quick_f(x.y); x.z = gelu_quick_f(x.z); x.w = gelu_quick_f(x.w); break;
        case MLSD_ACT_RELU: x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); break;
        default: break;
        }
        if (!p.act_post) { x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w; }
        if (p.C32) *reinterpret_cast<float4*>(p.C32 + (long)m * p.ldc32 + n) = x;
        if (p.C16) {
            f16x4 h = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
            *reinterpret_cast<f16x4*>(p.C16 + (long)m * p.ldc16 + n) = h;
        }
        cs.x += x.x; cs.y += x.y; cs.z += x.z; cs.w += x.w;
        cq.x += x.x * x.x; cq.y += x.y * x.y; cq.z += x.z * x.z; cq.w += x.w * x.w;
    }
    sh_s[rl][cgi] = cs; sh_q[rl][cgi] = cq;
    __syncthreads();
    if (rl == 0 && colok) {
        float4 a = sh_s[0][cgi], b = sh_q[0][cgi];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const float4 a2 = sh_s[r][cgi], b2 = sh_q[r][cgi];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        float* st = p.colstats + (long)blockIdx.y * 2 * p.N + n;
        *reinterpret_cast<float4*>(st) = a;
        *reinterpret_cast<float4*>(st + p.N) = b;
    }
}

// ---- split-K second pass that ENDS WITH THE LAYERNORM of the rows it finishes (round 4).  SD1.5 batch 1 is bound by its dispatch count; where the producer of a
// LayerNorm's input is a split-K launch, its reduce pass already holds every finished row: one block per row (N / 4 threads, a thread owns 4 consecutive columns)
// adds the slices in slice order and applies the epilogue exactly like splitk_reduce (bit-identical fp32 output), then takes the row's mean and centred sum of squares
// with two block reductions (fixed order) and writes fp16((v - mean) rstd gamma + beta): the LayerNorm launch (ggml_norm + mul + add, src/mlblock_nn.c:65-71)
// disappears.  No cross-block hand-off: nothing to time out.
__global__ __launch_bounds__(1024) void splitk_reduce_ln(const GemmP p, const float* __restrict__ ws, int nsplit)
{
    __shared__ float red[2][16];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const int n = tid * 4;
    const bool in = n < p.N;
    float4 v = make_float4(0, 0, 0, 0);
    if (in) {
        v = *reinterpret_cast<const float4*>(ws + (long)m * p.N + n);
        for (int z = 1; z < nsplit; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(ws + z * p.ws_stride + (long)m * p.N + n);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (p.biasm) { const float b = p.biasm[m]; v.x += b; v.y += b; v.z += b; v.w += b; }
        if (p.rowbias) {
            const float4 r = *reinterpret_cast<const float4*>(p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        float4 rs = make_float4(0, 0, 0, 0);
        if (p.resid) rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
        if (p.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
        switch (p.act) {
        case MLSD_ACT_SILU: v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); break;
        case MLSD_ACT_GELU: v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w); break;
        case MLSD_ACT_GELU_QUICK: v.x = gelu_quick_f(v.x); v.y = gelu_quick_f(v.y); v.z = gelu_quick_f(v.z); v.w = gelu_quick_f(v.w); break;
        case MLSD_ACT_RELU: v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); break;
        default: break;
        }
        if (!p.act_post) { v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w; }
        if (p.C32) *reinterpret_cast<float4*>(p.C32 + (long)m * p.ldc32 + n) = v;
    }
    // LayerNorm of the finished row: mean, then the centred sum of squares (two passes over the registers: no cancellation), waves combined in wave order
    float s = in ? (v.x + v.y) + (v.z + v.w) : 0.f;
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < nw; ++w) tot += red[0][w];
    const float mean = tot / (float)p.N;
    float q = 0.f;
    if (in) { const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean; q = (a * a + b * b) + (c * c + d * d); }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    float tq = 0.f;
    for (int w = 0; w < nw; ++w) tq += red[1][w];
    const float rstd = 1.0f / sqrtf(tq / (float)p.N + p.ln_eps);
    if (in) {
        const float4 g = *reinterpret_cast<const float4*>(p.ln_g + n), b = *reinterpret_cast<const float4*>(p.ln_b + n);
        f16x4 h = {(_Float16)((v.x - mean) * rstd * g.x + b.x), (_Float16)((v.y - mean) * rstd * g.y + b.y),
                   (_Float16)((v.z - mean) * rstd * g.z + b.z), (_Float16)((v.w - mean) * rstd * g.w + b.w)};
        *reinterpret_cast<f16x4*>(p.ln_y + (long)m * p.ldln + n) = h;
    }
}

#ifdef MLSD_GEMM_EXPERIMENTS
// ---- split-K second pass that ends with the GROUPNORM (+ SiLU) of what it finishes (round 4): one block per (image, group), the slab of HW rows x N / G columns in
// registers (<= GNR_MAXI float4 per thread: the sizes for which the one-dispatch GroupNorm of norm.hip wins).  Slices added in slice order + epilogue as splitk_reduce
// (bit-identical fp32 output), then mean / centred variance of the slab by two block reductions, normalise, affine, SiLU, fp16.  The GroupNorm launch that follows a
// split-K convolution in every resnet of SD1.5's 8x8 / 16x16 levels (ggml_group_norm + mul + add + silu, src/mlblock_nn.c:86-99,135-136,146-147) disappears.
constexpr int GNR_MAXI = 10;
__global__ __launch_bounds__(256) void splitk_reduce_gn(const GemmP p, const float* __restrict__ ws, int nsplit)
{
    __shared__ float red[8];
    const int g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = p.N / p.gn_G, Q4 = cg >> 2, items = p.gn_hw * Q4, c0 = g * cg;
    const int NI = (items + 255) >> 8;
    float4 v[GNR_MAXI];
    long off[GNR_MAXI];
#pragma unroll
    for (int i = 0; i < GNR_MAXI; ++i) {
        const int it = min(tid + i * 256, items - 1);
        const int pix = it / Q4;
        off[i] = ((long)img * p.gn_hw + pix) * p.N + c0 + 4 * (it - pix * Q4);      // element offset inside a slice ([M][N])
        v[i] = make_float4(0, 0, 0, 0);
    }
    for (int z = 0; z < nsplit; ++z) {          // slice order; the (up to) GNR_MAXI loads of a slice are in flight together
        float4 t[GNR_MAXI];
#pragma unroll
        for (int i = 0; i < GNR_MAXI; ++i) if (i < NI) t[i] = *reinterpret_cast<const float4*>(ws + z * p.ws_stride + off[i]);
#pragma unroll
        for (int i = 0; i < GNR_MAXI; ++i) if (i < NI) {
            if (z == 0) v[i] = t[i];
            else { v[i].x += t[i].x; v[i].y += t[i].y; v[i].z += t[i].z; v[i].w += t[i].w; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GNR_MAXI; ++i) {
        const int it = tid + i * 256;
        if (i < NI && it < items) {
            const int pix = it / Q4, n = c0 + 4 * (it - pix * Q4), m = img * p.gn_hw + pix;
            float4 x = v[i];
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w; }
            if (p.biasm) { const float b = p.biasm[m]; x.x += b; x.y += b; x.z += b; x.w += b; }
            if (p.rowbias) {
                const float4 r = *reinterpret_cast<const float4*>(p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb + n);
                x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
            }
            float4 rs = make_float4(0, 0, 0, 0);
            if (p.resid) rs = *reinterpret_cast<const float4*>(p.resid + (long)m * p.ldr + n);
            if (p.act_post) { x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w; }
            switch (p.act) {
            case MLSD_ACT_SILU: x.x = silu_f(x.x); x.y = silu_f(x.y); x.z = silu_f(x.z); x.w = silu_f(x.w); break;
            case MLSD_ACT_GELU: x.x = gelu_tanh_f(x.x); x.y = gelu_tanh_f(x.y); x.z = gelu_tanh_f(x.z); x.w = gelu_tanh_f(x.w); break;
            case MLSD_ACT_GELU_QUICK: x.x = gelu_quick_f(x.x); x.y = gelu_quick_f(x.y); x.z = gelu_quick_f(x.z); x.w = gelu_quick_f(x.w); break;
            case MLSD_ACT_RELU: x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); break;
            default: break;
            }
            if (!p.act_post) { x.x += rs.x; x.y += rs.y; x.z += rs.z; x.w += rs.w; }
            if (p.C32) *reinterpret_cast<float4*>(p.C32 + (long)m * p.ldc32 + n) = x;
            v[i] = x;
            s += (x.x + x.y) + (x.z + x.w);
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float cnt = (float)cg * (float)p.gn_hw;
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / cnt;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < GNR_MAXI; ++i)
        if (i < NI && tid + i * 256 < items) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / cnt + p.ln_eps);
#pragma unroll
    for (int i = 0; i < GNR_MAXI; ++i) {
        const int it = tid + i * 256;
        if (i < NI && it < items) {
            const int pix = it / Q4, n = c0 + 4 * (it - pix * Q4);
            const long m = (long)img * p.gn_hw + pix;
            const float4 ga = *reinterpret_cast<const float4*>(p.ln_g + n), be = *reinterpret_cast<const float4*>(p.ln_b + n);
            float y0 = (v[i].x - mean) * rstd * ga.x + be.x, y1 = (v[i].y - mean) * rstd * ga.y + be.y;
            float y2 = (v[i].z - mean) * rstd * ga.z + be.z, y3 = (v[i].w - mean) * rstd * ga.w + be.w;
            if (p.gn_silu) { y0 = silu_f(y0); y1 = silu_f(y1); y2 = silu_f(y2); y3 = silu_f(y3); }
            f16x4 h = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
            *reinterpret_cast<f16x4*>(p.ln_y + m * p.ldln + n) = h;
        }
    }
}
#endif  // MLSD_GEMM_EXPERIMENTS (GroupNorm in the reduce pass)

// number of K slices mlsd_gemm will actually use for these args (1 = no split)
int splitk_slices(const mlsd_gemm_args* a, int BK, int* kt_per_out)
{
    const int nkt = (a->K + BK - 1) / BK;
    int s = a->ksplit;
    if (s <= 1 || a->act == MLSD_ACT_GEGLU || !a->ws) { if (kt_per_out) *kt_per_out = nkt; return 1; }
    if (s > nkt) s = nkt;
    const int per = (nkt + s - 1) / s;
    if (kt_per_out) *kt_per_out = per;
    return (nkt + per - 1) / per;
}

int g_gemm_ncu = 256;   // CUs a persistent launch may occupy (mlsd_gemm_set_cus: half-chip partitions run 128-block grids)
int g_gemm_dbg = 0;
unsigned long long* g_gemm_tbuf = nullptr;   // device buffer of 8 stamps per block (mlsd_gemm_set_trace)
int g_gemm_panel = 8;   // tile-order panel width in tiles (0: row-major)
int g_gemm_epi = 0;    // 0: wide LDS-transposed epilogue when shapes allow (default)  1: scalar epilogue
// split-K reduced inside the launch (needs the caller's ticket counters).  OFF by default: measured SLOWER than the two-launch form on every
// SD1.5-sized problem (profiles/r3_gemm_splitk_inline.txt: 512x1280x5120 k/6 25.2 -> 29.0 us, 128x1280x5120 k/8 16.9 -> 20.3 us): the block that
// finishes a tile last reads all of the tile's slabs alone, at one CU's load bandwidth, where the second launch spreads the same bytes over the chip.
int g_gemm_sk_inline = 0;
// split-K whose slices are added IN the launch by all blocks of a tile together (gemm_kernel PAR, round 4): on by default where the whole grid is resident
// EXPERIMENTS builds only, and off unless MLSD_SPLITK_PAR=1 / mlsd_gemm_set_splitk_parallel(1): measured slower than the two-launch form (profiles/NOTES.md)
int g_gemm_sk_par = -1;
bool sk_par_on()
{
    if (g_gemm_sk_par < 0) { const char* e = getenv("MLSD_SPLITK_PAR"); g_gemm_sk_par = (e && *e && *e != '0') ? 1 : 0; }
    return g_gemm_sk_par != 0;
}

int device_cus();
// may this split-K launch add its slices itself (gemm_kernel PAR)?  Every block waits for the other slices of its tile, so the WHOLE grid must be resident at once:
// tiles x slices <= CUs x blocks per CU of the tile (64x128: 3, 128x128: 2; LDS-bound), on an unpartitioned device; counters: 2047 tiles at most
bool splitk_par_ok(const mlsd_gemm_args* a, int BM, int nsplit, long tiles)
{
#ifndef MLSD_GEMM_EXPERIMENTS
    return false;       // (the kernels are not in the product build)
#endif
    if (!sk_par_on() || !a->sk_flags || nsplit < 2 || tiles > 2047) return false;
    const int cus = device_cus();
    if (cus < g_gemm_ncu || g_gemm_ncu < 256) return false;
    return tiles * nsplit <= (long)cus * (BM == 64 ? 3 : 2);
}

// Rows per column-statistics block a launch of the GENERAL tiles would write for `a` (0: none).  Non-split: the wide epilogue of gemm_kernel sums per wave (WM rows);
// split-K: splitk_reduce_stats (32 rows).  Needs the wide epilogue (alignment), an fp32 output, whole 64-column slabs per wave (even TN) and no fused norm.
int general_stats_rows(const mlsd_gemm_args* a, int WM, int TN, int BK)
{
    if (!a->colstats || ((uintptr_t)a->colstats & 15) || !a->C32 || a->act == MLSD_ACT_GEGLU || a->ln_y16 || a->gn_y16 || (TN & 1) || (a->N & 3)) return 0;
    const bool vec = !(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15) && (!a->C16 || (!(a->ldc16 & 3) && !((uintptr_t)a->C16 & 7))) &&
                     (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) && (!a->bias || !((uintptr_t)a->bias & 15)) &&
                     (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15))) && g_gemm_epi != 1;
    if (!vec || g_gemm_sk_inline) return 0;
    const int nsplit = splitk_slices(a, BK, nullptr);
    if (nsplit > 1) {
#ifdef MLSD_GEMM_EXPERIMENTS
        if (sk_par_on()) return 0;          // (the slices may be added inside the launch: no reduce pass)
#endif
        return 32;
    }
    return WM;
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int NSTAGE, bool REG = false>
int launch(const mlsd_gemm_args* a, hipStream_t st)
{
    GemmP p;
    p.A = (const _Float16*)a->A; p.B = (const _Float16*)a->W_;
    p.lda = a->lda; p.ldb = a->ldb; p.M = a->M; p.N = a->N; p.K = a->K;
    p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.OH = a->OH; p.OW = a->OW; p.KH = a->KH; p.KW = a->KW;
    p.stride = a->stride; p.pad = a->pad; p.ups = a->upsample; p.korder = 0;
    p.bias = a->bias; p.biasm = a->bias_m; p.act_post = a->act_after_resid; p.rowbias = a->rowbias; p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
    p.ldrb = a->ldrb; p.resid = a->resid; p.ldr = a->ldr; p.act = a->act;
    p.C32 = a->C32; p.ldc32 = a->ldc32; p.C16 = (_Float16*)a->C16; p.ldc16 = a->ldc16;
    p.nbm = (a->M + BM - 1) / BM; p.nbn = (a->N + BN - 1) / BN;
    {
        const int nout = a->act == MLSD_ACT_GEGLU ? a->N / 2 : a->N;
        p.vec = !(nout & 3) && !(a->N & 3) && (!a->C32 || (!(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15))) &&
                (!a->C16 || (!(a->ldc16 & 3) && !((uintptr_t)a->C16 & 7))) && (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) &&
                (!a->bias || !((uintptr_t)a->bias & 15)) && (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15))) && g_gemm_epi != 1;
    }
    p.dbg = g_gemm_dbg; p.tbuf = nullptr; p.colstats = nullptr; p.sk_L = 0; p.sk_ws = nullptr; p.sk_flag = nullptr;
    p.gw = g_gemm_panel;
    int kt_per;
    const int nsplit = p.vec ? splitk_slices(a, BK, &kt_per) : 1;
    p.kt_per = nsplit > 1 ? kt_per : (a->K + BK - 1) / BK;
    p.ws_stride = 0;
    // column statistics for a consuming GroupNorm (general_stats_rows: the same conditions mlsd_gemm_colstats_rows promised the planner): one K slice -> this kernel's
    // epilogue (blocks of WM rows); split-K -> the reduce pass (blocks of 32 rows)
    const int st_rows = a->colstats ? general_stats_rows(a, BM / WAVES_M, (BN / WAVES_N) / 32, BK) : 0;
    if (st_rows > 0 && nsplit == 1) p.colstats = a->colstats;
    GemmP pe = p;                                  // the epilogue as requested (second pass of a split-K launch)
    if (st_rows > 0 && nsplit > 1) pe.colstats = a->colstats;
    // reduced inside the launch: one counter per output tile (a->sk_flags, 4096 words, zero between launches), one slab per (tile, slice)
    const bool inl = nsplit > 1 && g_gemm_sk_inline && a->sk_flags && (long)p.nbm * p.nbn <= 4095 &&        /* (word 4095 is the sticky give-up indicator of the stream-K hand-offs) */
                     !((uintptr_t)a->ws & 15) &&
                     a->ws_bytes >= (size_t)nsplit * p.nbm * p.nbn * BM * BN * sizeof(float);
    if (inl) { p.sk_L = nsplit; p.sk_ws = (float*)a->ws; p.sk_flag = a->sk_flags; }
    else if (nsplit > 1) {
        const size_t need = (size_t)nsplit * a->M * a->N * sizeof(float);
        if (a->ws_bytes < need || ((uintptr_t)a->ws & 15))
            return mlsd_set_error(-1, "mlsd_gemm: split-K workspace too small or misaligned (%zu < %zu)", (size_t)a->ws_bytes, need);
        p.bias = p.biasm = p.rowbias = p.resid = nullptr; p.act = 0; p.act_post = 0;
        p.C16 = nullptr; p.C32 = (float*)a->ws; p.ldc32 = a->N; p.ws_stride = (long)a->M * a->N;
        pe.ws_stride = p.ws_stride;
    }
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    constexpr size_t RING = (size_t)NSTAGE * (BM + BN) * BK * 2;
    constexpr size_t EPI = (size_t)WAVES_M * WAVES_N * 32 * 64 * 4;      // 8 KiB per wave
    constexpr size_t LDS = RING > EPI ? RING : EPI;
    const dim3 grid(p.nbm * p.nbn, nsplit), block(THREADS);
    // ---- slices added in the launch by the blocks of the tile together (gemm_kernel PAR): the small tiles of the split-K launches, whole grid resident
#ifdef MLSD_GEMM_EXPERIMENTS   /* measured SLOWER than the second launch (SD1.5 b1 evaluation 7.17 -> 7.76 ms, profiles/NOTES.md): a dispatch boundary is the cheaper grid barrier */
    if constexpr ((BM == 64 || BM == 128) && BN == 128 && BK == 64 && NSTAGE == 2 && !REG) {
        if (nsplit > 1 && !inl && splitk_par_ok(a, BM, nsplit, p.nbm * p.nbn)) {
            p.sk_flag = a->sk_flags;
            auto kfn = a->conv ? gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, true, NSTAGE, 0, false, true> : gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, false, NSTAGE, 0, false, true>;
            if (LDS > 65536) MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            hipLaunchKernelGGL(kfn, grid, block, LDS, st, p, pe);
            return mlsd_check_launch("gemm_kernel(split-K, reduced in the launch)");
        }
    }
#endif
    auto go = [&](auto kfn) -> int {
        if (LDS > 65536) MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        hipLaunchKernelGGL(kfn, grid, block, LDS, st, p, pe);
        if (nsplit > 1 && !inl) {
            const long n = (long)a->M * (a->N >> 2);
#ifdef MLSD_GEMM_EXPERIMENTS
            if (a->gn_y16) {       // the reduce pass ends with the GroupNorm of its (image, group) slabs (mlsd_gemm_gn_fused; mlsd_gemm checked that this launch qualifies)
                GemmP pl = pe;
                pl.ln_g = a->gn_gamma; pl.ln_b = a->gn_beta; pl.ln_eps = a->gn_eps; pl.ln_y = (_Float16*)a->gn_y16; pl.ldln = a->gn_ldy;
                pl.gn_G = a->gn_groups; pl.gn_hw = a->gn_hw; pl.gn_silu = a->gn_silu;
                hipLaunchKernelGGL(splitk_reduce_gn, dim3((unsigned)a->gn_groups, (unsigned)(a->M / a->gn_hw)), dim3(256), 0, st, pl, (const float*)a->ws, nsplit);
            } else
#endif
            if (a->ln_y16) {       // the reduce pass ends with the LayerNorm of its rows (mlsd_gemm_ln_fused == 2; mlsd_gemm checked that this launch qualifies)
                GemmP pl = pe;
                pl.ln_g = a->ln_gamma; pl.ln_b = a->ln_beta; pl.ln_eps = a->ln_eps; pl.ln_y = (_Float16*)a->ln_y16; pl.ldln = a->ldln;
                hipLaunchKernelGGL(splitk_reduce_ln, dim3((unsigned)a->M), dim3((unsigned)(((a->N >> 2) + 63) / 64 * 64)), 0, st, pl, (const float*)a->ws, nsplit);
            } else if (pe.colstats)
                hipLaunchKernelGGL(splitk_reduce_stats, dim3((unsigned)(((a->N >> 2) + 63) / 64), (unsigned)((a->M + 31) / 32)), dim3(256), 0, st, pe, (const float*)a->ws, nsplit);
            else
            hipLaunchKernelGGL(splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pe, (const float*)a->ws, nsplit);
        }
        return mlsd_check_launch("gemm_kernel");
    };
#ifdef MLSD_GEMM_EXPERIMENTS   /* timing-only builds of the loop: no refills (1) / no MFMA (2); see DESIGN.md */
    if (!a->conv && g_gemm_dbg == 1) return go(gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, false, NSTAGE, 1, REG>);
    if (!a->conv && g_gemm_dbg == 2) return go(gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, false, NSTAGE, 2, REG>);
#endif
    if constexpr (((BN / WAVES_N) / 32) % 2 == 0 && !REG) {       // the statistics builds (whole 64-column wave slabs)
        if (p.colstats) return a->conv ? go(gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, true, NSTAGE, 0, false, false, true>) : go(gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, false, NSTAGE, 0, false, false, true>);
    }
    p.colstats = nullptr;
    return a->conv ? go(gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, true, NSTAGE, 0, REG>) : go(gemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, false, NSTAGE, 0, REG>);
}

// CUs of the current device (cached per device id).  The in-launch hand-offs (stream-K, LayerNorm statistics) wait for blocks of the SAME launch: all
// blocks of the persistent grid must be resident at once, i.e. the device must have at least g_gemm_ncu CUs (a partitioned or masked device does not).
int device_cus()
{
    static thread_local int cached_dev = -1, cached = 0;     // (per thread: MLCtx objects are used from several threads)
    int dev = 0;
    if (mlsd_runtime_is_dry() || hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev != cached_dev) {
        hipDeviceProp_t pr;
        cached = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 0;
        cached_dev = dev;
    }
    return cached;
}

// problems the ping-pong kernels take (gemm_pp.hpp); everything else asked of variants 17 / 18 runs on the LDS-transposing
// tile of the same shape (9 / 16)
int g_gemm_korder = 0;     // experiment (EXPERIMENTS builds): slab-ordered K for the ping-pong convolutions, timing only -- see mlsd_gemm_set_korder
int pp_epilogue_kind(const mlsd_gemm_args* a, int BN);
bool pp_eligible(const mlsd_gemm_args* a, int BM, int BN)
{
    if ((a->K & 63) || a->K < 192 || (a->M % (BM / 2)) || (a->N % (BN / 4))) return false;
    if (a->conv && ((a->Cin & 63) || a->KH * a->KW > 9)) return false;   // a K tile must lie inside one filter tap
    if (a->conv && a->upsample) {       // nearest-2x upsampled source (round 5): stride 1, and the two epilogues the upsampling convolutions of the UNets / decoders use
        const int e = pp_epilogue_kind(a, BN);
        if (a->stride != 1 || a->pad > 16 || a->H > 16000 || a->W > 16000 || (e != 2 /* PP_EPI_F32 */ && e != 5 /* PP_EPI_F32_STATS */)) return false;
    }
    if (a->rowbias && ((a->rows_per_batch > 0 ? a->rows_per_batch : 1) % BM)) return false;
    if (a->act == MLSD_ACT_GEGLU && BN != 256) return false;
    const int nout = a->act == MLSD_ACT_GEGLU ? a->N / 2 : a->N;
    return !(nout & 3) && (!a->C32 || (!(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15))) && (!a->C16 || (!(a->ldc16 & 3) && !((uintptr_t)a->C16 & 7))) &&
           (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) && (!a->bias || !((uintptr_t)a->bias & 15)) &&
           (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15))) && g_gemm_epi != 1;
}

// launches that can END with the LayerNorm of their output (gemm_pp.hpp *_LN): linear, 128 x 320 ping-pong tile made of WHOLE tiles, all of them resident at
// once (single round: the tiles of a row block exchange their partial row statistics inside the launch), fp32 output (+ residual), no activation / statistics / row bias
bool ln_eligible(const mlsd_gemm_args* a)
{
    if (!a->ln_y16 || !a->ln_gamma || !a->ln_beta || !a->ln_ws || !a->ln_cnt || a->conv || a->colstats || a->rowbias || a->bias_m || a->act != MLSD_ACT_NONE) return false;
    if (!a->C32 || a->C16 || (a->M % 128) || (a->N % 320) || (a->ldln & 7) || ((uintptr_t)a->ln_y16 & 15) || ((uintptr_t)a->ln_gamma & 15) || ((uintptr_t)a->ln_beta & 15)) return false;
    // the tiles of a row block must run at the same time.  One round (tiles <= blocks): all of them do.  Several rounds on the full chip: the tile order hands the nbn
    // tiles of a row block to blocks b, b + 8, ... of ONE round when whole rounds of 256 tiles are made of whole groups of 8 nbn tiles (nbn = 1, 2, 4)
    const long nbn = a->N / 320, tiles = (long)(a->M / 128) * nbn;
    const bool one_round = tiles <= g_gemm_ncu;
    const bool whole_rounds = g_gemm_ncu == 256 && !(tiles % 256) && (nbn == 1 || nbn == 2 || nbn == 4);
    return (one_round || whole_rounds) && nbn <= 8 && a->ln_slot >= 0 && a->ln_slot < 8191 && pp_eligible(a, 128, 320) && device_cus() >= g_gemm_ncu;      // (<= 8 partner tiles per row: two per lane group of the gathering wave)
}

// launches that END with the cross attention of the q they project (gemm_pp.hpp PP_EPI_XATTN): linear, whole 128 x 320 tiles (5 heads of 64), fp16 "output" that is never
// stored, K / V^T / output operands, a tile inside one image, at most 77 keys held in LDS.  MLSD_XATTN=0 switches the form off (the plan builder asks mlsd_gemm_xattn_fused).
int g_xattn_mode = -1;      // 0 = never, 1 = where it wins (the rule below), 2 = wherever the kernel takes the launch; -1: MLSD_XATTN from the environment (default 1); mlsd_gemm_set_xattn
int xattn_mode()
{
    if (g_xattn_mode < 0) { const char* e = getenv("MLSD_XATTN"); g_xattn_mode = (e && *e >= '0' && *e <= '2') ? *e - '0' : 1; }
    return g_xattn_mode;
}
bool xattn_eligible(const mlsd_gemm_args* a)
{
    if (!a->xa_k || !a->xa_vt || !a->xa_out || !xattn_mode() || a->conv || a->act != MLSD_ACT_NONE || a->C32 || a->resid || a->rowbias || a->bias_m || a->colstats) return false;
    if (a->ln_y16 || a->gn_y16 || (a->M % 128) || (a->N % 320) || (a->K & 63) || a->K < 192) return false;
    if (a->xa_Tq <= 0 || (a->xa_Tq % 128) || (a->M % a->xa_Tq) || a->xa_Tk < 1 || a->xa_Tk > 77) return false;
    // Only where the 128 x 320 tiles fill more than half of the CUs: below that the projection alone runs on the 128 x 160 two-per-CU kernel (twice the blocks) and the
    // unfused pair wins -- measured (tools/xattn_bench.py, profiles/r6_xattn_bench.txt): 8192 x 1280 (256 tiles) fused 40.8 us against 50.7; 4096 x 1280 (128 tiles) 35.6 against
    // 29.8; 2048 x 1280 33.3 against 26.3; in the plan SDXL b1 +1.2 % with the fused form everywhere.  (MLSD_XATTN=2 / mlsd_gemm_set_xattn(2): everywhere, for that A/B and the kernel tests.)
    if (xattn_mode() != 2 && (long)(a->M / 128) * (a->N / 320) <= g_gemm_ncu / 2) return false;
    if ((a->xa_ldk & 7) || (a->xa_ldo & 7) || ((uintptr_t)a->xa_k & 15) || ((uintptr_t)a->xa_vt & 15) || ((uintptr_t)a->xa_out & 15) || (a->bias && ((uintptr_t)a->bias & 15))) return false;
    return g_gemm_epi != 1 && !(g_gemm_dbg & 2);
}

// which epilogue body a ping-pong launch of these arguments uses (gemm_pp.hpp PP_EPI_*)
int pp_epilogue_kind(const mlsd_gemm_args* a, int BN)
{
    if (a->xa_k && BN == 320 && xattn_eligible(a)) return PP_EPI_XATTN;
    if ((g_gemm_dbg & 2) || a->bias_m) return PP_EPI_GENERIC;
    const bool c16_wide = a->C16 && !(a->ldc16 & 7) && !((uintptr_t)a->C16 & 15);     // the fp16 fast paths store 16 bytes per lane
    if (a->act == MLSD_ACT_NONE) {
        if (a->C16 && !a->C32 && !a->resid) return c16_wide ? PP_EPI_F16 : PP_EPI_GENERIC;
        if (a->C32 && !a->C16) {
            if (a->ln_y16 && ln_eligible(a)) return a->resid ? PP_EPI_F32_RES_LN : PP_EPI_F32_LN;
            const bool st = a->colstats != nullptr && !((uintptr_t)a->colstats & 15) && !(a->N & 3);
            return a->resid ? (st ? PP_EPI_F32_RES_STATS : PP_EPI_F32_RES) : (st ? PP_EPI_F32_STATS : PP_EPI_F32);
        }
    } else if (a->act == MLSD_ACT_GEGLU && BN == 256 && c16_wide && !a->C32 && !a->resid && !a->conv) return PP_EPI_GEGLU16;
    return PP_EPI_GENERIC;
}

// launcher of the ping-pong kernels (gemm_pp.hpp): same argument handling as launch<>
// stream-K needs: the ping-pong conditions, a workspace of one fp32 slab per block + a zeroed flag word per block (cleared again by
// their consumers), at least 4 K-tile units per block and a tile count the blocks do not divide (otherwise nothing is gained)
// K tiles per stream-K block.  The even share ceil(units / blocks) would cut every output tile at different K positions, so the fp32 summation
// order of a row would depend on which tile it lands in -- and an image's bits on its batch slot (tests/test_determinism_gpu.py).  The share is
// therefore rounded UP to a divisor of the tile's K-tile count: every tile is cut at the same positions (a split-K whose slices are dealt over the
// persistent blocks), at the price of some idle blocks (96 tiles x 90 K tiles on 256 blocks: 45 per block = 192 busy blocks instead of 34 on all).
int sk_share(long ntiles, int nkt, int ncu)
{
    const long even = (ntiles * nkt + ncu - 1) / ncu;
    for (int L = (int)(even < 4 ? 4 : even); L < nkt; ++L) if (nkt % L == 0) return L;      // (>= 4 K tiles per block)
    return nkt;
}

bool sk_eligible(const mlsd_gemm_args* a, int BM, int BN, bool ignore_stats = false)
{
    if (!pp_eligible(a, BM, BN) || !a->ws || !a->sk_flags || ((uintptr_t)a->ws & 15) || device_cus() < g_gemm_ncu || (a->conv && a->upsample)) return false;
    if (a->conv && a->colstats && !ignore_stats) return false;      // (the conv builds with the statistics epilogue do not fit the register budget beside the hand-off code)
    const long tiles = (long)((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN), nkt = a->K / 64;
    const long L = sk_share(tiles, (int)nkt, g_gemm_ncu);
    return nkt >= 3 && L >= 4 && L < nkt && a->ws_bytes >= (size_t)g_gemm_ncu * BM * BN * sizeof(float);    // (L == nkt: nothing to split)
}

template <int BM, int BN, int CB0, int CB1, bool RESBATCH, bool SK = false, int NPH = 4, int SCH = 0>
int launch_pp(const mlsd_gemm_args* a, hipStream_t st)
{
    constexpr int BK = 64;
    GemmP p;
    p.A = (const _Float16*)a->A; p.B = (const _Float16*)a->W_;
    p.lda = a->lda; p.ldb = a->ldb; p.M = a->M; p.N = a->N; p.K = a->K;
    p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.OH = a->OH; p.OW = a->OW; p.KH = a->KH; p.KW = a->KW;
    p.stride = a->stride; p.pad = a->pad; p.ups = a->upsample; p.korder = g_gemm_korder;
    p.bias = a->bias; p.biasm = a->bias_m; p.act_post = a->act_after_resid; p.rowbias = a->rowbias; p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
    p.ldrb = a->ldrb; p.resid = a->resid; p.ldr = a->ldr; p.act = a->act;
    p.C32 = a->C32; p.ldc32 = a->ldc32; p.C16 = (_Float16*)a->C16; p.ldc16 = a->ldc16;
    p.nbm = (a->M + BM - 1) / BM; p.nbn = (a->N + BN - 1) / BN;
    p.vec = 1;                                             // pp_eligible() checked the alignment
    p.dbg = g_gemm_dbg; p.gw = g_gemm_panel; p.tbuf = g_gemm_tbuf; p.colstats = nullptr;
    p.kt_per = (a->K + BK - 1) / BK; p.ws_stride = 0;      // no split-K on these tiles
    constexpr size_t LDS = 2 * (size_t)(BM + BN) * BK * 2; // the ring; the epilogue needs no LDS
    const int ntiles = p.nbm * p.nbn;
    p.sk_L = 0; p.sk_ws = nullptr; p.sk_flag = nullptr;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_y = (_Float16*)a->ln_y16; p.ldln = a->ldln; p.ln_ws = a->ln_ws; p.ln_cnt = a->ln_cnt; p.ln_slot = a->ln_slot;
    if constexpr (SK) {
        p.sk_L = sk_share(ntiles, a->K / BK, g_gemm_ncu);
        p.sk_ws = (float*)a->ws; p.sk_flag = a->sk_flags;
    }
    const dim3 grid(SK ? g_gemm_ncu : (ntiles < g_gemm_ncu ? ntiles : g_gemm_ncu)), block(512);   // persistent: one block per CU walks tiles b, b+G, ... (stream-K: K-tile units)
    auto go = [&](auto kfn) -> int {
        MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        hipLaunchKernelGGL(kfn, grid, block, LDS, st, p);
        return mlsd_check_launch("gemm_pp_kernel");
    };
    auto go_ln = [&](auto kfn) -> int {              // + 4 KB beyond the ring for the wave columns' partial row statistics + 8 KB for the partner tiles' (<= 8 per row)
        MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS + 4096 + 8192)));
        hipLaunchKernelGGL(kfn, grid, block, LDS + 4096 + 8192, st, p);
        return mlsd_check_launch("gemm_pp_kernel(+LN)");
    };
    auto go_xa = [&](auto kfn) -> int {              // cross attention at the end: ONE tile per block (the epilogue re-uses the ring's LDS), K + q images = 138 KB
        constexpr int XLDS = 57344 + 128 * 640;
        p.xa_k = (const _Float16*)a->xa_k; p.xa_vt = (const _Float16*)a->xa_vt; p.xa_out = (_Float16*)a->xa_out; p.xa_ldk = a->xa_ldk; p.xa_ldo = a->xa_ldo;
        p.xa_Tq = a->xa_Tq; p.xa_Tk = a->xa_Tk; p.xa_sc = 1.4426950408889634f * 0.125f;      /* log2(e) / sqrt(64) */
        MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS));
        hipLaunchKernelGGL(kfn, dim3(ntiles), block, XLDS, st, p);
        return mlsd_check_launch("gemm_pp_kernel(+attention)");
    };
    // the epilogue the kernel is built with (gemm_pp.hpp): the bulk launches of the UNet / VAE have no activation in the GEMM
    const int epi = pp_epilogue_kind(a, BN);
    p.colstats = (epi == PP_EPI_F32_STATS || epi == PP_EPI_F32_RES_STATS) ? a->colstats : nullptr;
    if constexpr (SK) {       // the stream-K builds: the fp32 epilogues of the long-K convs / feed-forward outputs, fp16 for the fused projections
        if (a->conv) {
            switch (epi) {
            case PP_EPI_F32: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32, true>);
            case PP_EPI_F32_RES: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32_RES, true>);
            case PP_EPI_F32_STATS: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32_STATS, true>);
            case PP_EPI_F32_RES_STATS: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32_RES_STATS, true>);
            default: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_GENERIC, true>);
            }
        }
        switch (epi) {
        case PP_EPI_F16: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F16, true>);
        case PP_EPI_F32: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32, true>);
        case PP_EPI_F32_RES: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32_RES, true>);
        default: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_GENERIC, true>);
        }
    }
    if (a->conv && a->upsample) {       // (pp_eligible admits these two epilogues only)
        if (epi == PP_EPI_F32_STATS) return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, 2, PP_EPI_F32_STATS, false, NPH, SCH>);
        if (epi == PP_EPI_F32) return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, 2, PP_EPI_F32, false, NPH, SCH>);
        return mlsd_set_error(-1, "mlsd_gemm: ping-pong tile: upsampled convolution with epilogue %d", epi);
    }
    if (a->conv) {
        switch (epi) {
        case PP_EPI_F32: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32, false, NPH, SCH>);
        case PP_EPI_F32_RES: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32_RES, false, NPH, SCH>);
        case PP_EPI_F32_STATS: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32_STATS, false, NPH, SCH>);
        case PP_EPI_F32_RES_STATS: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_F32_RES_STATS, false, NPH, SCH>);
        default: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, true, PP_EPI_GENERIC, false, NPH, SCH>);
        }
    }
    switch (epi) {
    case PP_EPI_F16: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F16, false, NPH, SCH>);
    case PP_EPI_F32: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32, false, NPH, SCH>);
    case PP_EPI_F32_RES: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32_RES, false, NPH, SCH>);
    case PP_EPI_F32_STATS: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32_STATS, false, NPH, SCH>);
    case PP_EPI_F32_RES_STATS: return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32_RES_STATS, false, NPH, SCH>);
    case PP_EPI_F32_LN:
        if constexpr (BM == 128 && BN == 320 && NPH == 4 && SCH == 0) return go_ln(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32_LN, false, 4, 0>);
        break;
    case PP_EPI_F32_RES_LN:
        if constexpr (BM == 128 && BN == 320 && NPH == 4 && SCH == 0) return go_ln(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_F32_RES_LN, false, 4, 0>);
        break;
    case PP_EPI_GEGLU16:
        if constexpr (BN == 256) return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_GEGLU16, false, NPH, SCH>);
        break;
    case PP_EPI_XATTN:
        if constexpr (BM == 128 && BN == 320 && NPH == 4 && SCH == 0) return go_xa(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_XATTN, false, 4, 0>);
        return mlsd_set_error(-1, "mlsd_gemm: cross attention at the end of a projection runs on the four-phase 128x320 tile only");
    default: break;
    }
    return go(gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, false, PP_EPI_GENERIC, false, NPH, SCH>);
}

// ---- skinny-M weight-streaming kernel (gemm_skinny.hpp, tile variant 29): M <= 128, K in whole 64-wide steps, conv taps of whole steps; the K slices go to the
// split-K workspace and splitk_reduce (declared above) adds them and applies the epilogue -- also for ONE slice
bool skinny_eligible(const mlsd_gemm_args* a)
{
    if (a->M > 128 || a->act == MLSD_ACT_GEGLU || (a->K & 63) || a->K < 128 || (a->N & 3) || !a->ws || ((uintptr_t)a->ws & 15)) return false;
    if (a->conv && (a->upsample || (a->Cin & 63) || a->KH * a->KW > 9)) return false;
    return (!a->C32 || (!(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15))) && (!a->C16 || (!(a->ldc16 & 3) && !((uintptr_t)a->C16 & 7))) &&
           (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) && (!a->bias || !((uintptr_t)a->bias & 15)) &&
           (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15)));
}

int skinny_slices(const mlsd_gemm_args* a, int* kt_per_out)
{
    const int nkt = a->K / 64;
    int s = a->ksplit < 1 ? 1 : a->ksplit;
    if (s > nkt / 2) s = nkt / 2 > 0 ? nkt / 2 : 1;          // >= 2 K steps per slice
    const int maxper = SKINNY_MAXG * 4;                      // (the kernel is straight-line code for up to SKINNY_MAXG groups of PF + 1 = 4 K steps)
    if ((nkt + s - 1) / s > maxper) s = (nkt + maxper - 1) / maxper;
    const int per = (nkt + s - 1) / s;
    if (kt_per_out) *kt_per_out = per;
    return (nkt + per - 1) / per;
}

int launch_skinny(const mlsd_gemm_args* a, hipStream_t st)
{
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.A = (const _Float16*)a->A; p.B = (const _Float16*)a->W_;
    p.lda = a->lda; p.ldb = a->ldb; p.M = a->M; p.N = a->N; p.K = a->K;
    p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.OH = a->OH; p.OW = a->OW; p.KH = a->KH; p.KW = a->KW;
    p.stride = a->stride; p.pad = a->pad; p.ups = 0;
    p.rows_per_batch = 1; p.vec = 1; p.gw = 0;
    int kt_per;
    const int nsplit = skinny_slices(a, &kt_per);
    const size_t need = (size_t)nsplit * a->M * a->N * sizeof(float);
    if (a->ws_bytes < need) return mlsd_set_error(-1, "mlsd_gemm: skinny-M workspace too small (%zu < %zu)", (size_t)a->ws_bytes, need);
    p.kt_per = kt_per; p.C32 = (float*)a->ws; p.ldc32 = a->N; p.ws_stride = (long)a->M * a->N;
    GemmP pe = p;                                  // the epilogue as requested: applied by splitk_reduce
    pe.bias = a->bias; pe.biasm = a->bias_m; pe.act_post = a->act_after_resid; pe.rowbias = a->rowbias; pe.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
    pe.ldrb = a->ldrb; pe.resid = a->resid; pe.ldr = a->ldr; pe.act = a->act;
    pe.C32 = a->C32; pe.ldc32 = a->ldc32; pe.C16 = (_Float16*)a->C16; pe.ldc16 = a->ldc16;
    // K steps in flight per thread.  Measured (profiles/r4_gemm_skinny_ablation.txt, 128x1280x11520 conv, cold weights): PF = 7 (128 KB of LDS, one block per CU) 30.2 us,
    // PF = 3 (64 KB, two blocks per CU: one block's barrier / LDS / MFMA chain overlaps the other's) 23.9 us; the weight stream alone 19.2 / 16.1 us
    constexpr int PF = 3;
    const dim3 grid((a->N + 63) / 64, nsplit), block(256);
    auto go = [&](auto kfn, int rows, int pf = PF) -> int {
        const size_t lds = (size_t)(pf + 1) * rows * 128;
        static thread_local const void* attr_done[16]; static thread_local int n_done = 0;      // (one host call per kernel, not per launch)
        bool seen = false;
        for (int i = 0; i < n_done; ++i) seen |= attr_done[i] == (const void*)kfn;
        if (!seen) {
            MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (n_done < 16) attr_done[n_done++] = (const void*)kfn;
        }
        hipLaunchKernelGGL(kfn, grid, block, lds, st, p);
        const long n = (long)a->M * (a->N >> 2);
        hipLaunchKernelGGL(splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pe, (const float*)a->ws, nsplit);
        return mlsd_check_launch("gemm_skinny_kernel");
    };
#ifdef MLSD_GEMM_EXPERIMENTS   /* timing-only builds (mlsd_gemm_set_debug): 16 = the weight stream alone with 7 K steps in flight (128 KB of LDS), 32 = that depth with the arithmetic, 48 = the stream alone at the product depth */
    if (g_gemm_dbg == 16) return a->conv ? go(gemm_skinny_kernel<true, 8, 7, 1>, 128, 7) : go(gemm_skinny_kernel<false, 8, 7, 1>, 128, 7);
    if (g_gemm_dbg == 32) return a->conv ? go(gemm_skinny_kernel<true, 8, 7>, 128, 7) : go(gemm_skinny_kernel<false, 8, 7>, 128, 7);
    if (g_gemm_dbg == 48) return a->conv ? go(gemm_skinny_kernel<true, 8, 3, 1>, 128, 3) : go(gemm_skinny_kernel<false, 8, 3, 1>, 128, 3);
#endif
    if (a->M <= 16) return a->conv ? go(gemm_skinny_kernel<true, 1, PF>, 32) : go(gemm_skinny_kernel<false, 1, PF>, 32);
    if (a->M <= 64) return a->conv ? go(gemm_skinny_kernel<true, 4, PF>, 64) : go(gemm_skinny_kernel<false, 4, PF>, 64);
    return a->conv ? go(gemm_skinny_kernel<true, 8, PF>, 128) : go(gemm_skinny_kernel<false, 8, PF>, 128);
}

int g_gemm_variant = -1;   // >=0: forced tile variant (benchmarking)
#ifdef MLSD_GEMM_EXPERIMENTS
extern "C" int mlsd_gemm_w4_eligible(const mlsd_gemm_args* a, int tile);    // gemm_w4.hip (tile 0: 256 x 256, 1: 128 x 320)
extern "C" int mlsd_gemm_w4(const mlsd_gemm_args* a, int tile, void* stream, int ncu);
#else   /* the one-wave-per-SIMD tiles (variants 26 / 27) are not in the product build: the variants run as the ping-pong tile of the same shape */
static int mlsd_gemm_w4_eligible(const mlsd_gemm_args*, int) { return 0; }
static int mlsd_gemm_w4(const mlsd_gemm_args*, int, void*, int) { return -1; }
#endif

extern "C" int mlsd_gemm_tt_eligible(const mlsd_gemm_args* a, int ncu);     // gemm_tt.hip: 128 x 160 tile, 4 waves, two blocks per CU (variant 30)
extern "C" int mlsd_gemm_tt(const mlsd_gemm_args* a, void* stream, int ncu);
extern "C" int mlsd_conv_smalln_eligible(const mlsd_gemm_args* a);                // conv_smalln.hip: 3x3 convolutions with Cout <= 16 as a streaming op (variant 31)
extern "C" int mlsd_conv_smalln(const mlsd_gemm_args* a, void* stream);
// CUs the 128 x 160 kernel may count on for launches whose tiles wait for each other (LayerNorm endings: all partner tiles resident together): the stream's budget, and not
// more than the device has (a partitioned device); the dry runtime has no device and plans for the full part
static int tt_ncu() { if (mlsd_runtime_is_dry()) return g_gemm_ncu; const int c = device_cus(); return c < g_gemm_ncu ? c : g_gemm_ncu; }

struct Variant { const char* name; int bm, bn, slots; };
const Variant kVariants[] = {
    {"128x128x64s2", 128, 128, 512},   // 0: 64 KB ring, 2 blocks/CU
    {"64x128x64s2", 64, 128, 768},     // 1: small M
    {"128x128x32s4", 128, 128, 512},   // 2: 4-deep ring of 16 KB stages, 2 blocks/CU
    {"256x128x64s2", 256, 128, 256},   // 3: 8 waves, 96 KB, 1 block/CU
    {"256x128x32s3", 256, 128, 512},   // 4: 8 waves, 3-deep ring of 24 KB stages (72 KB), 2 blocks/CU
    {"128x128x64s3", 128, 128, 256},   // 5: 3-deep ring, 96 KB, 1 block/CU
    {"256x256x32s3", 256, 256, 256},   // 6: 8 waves (wave tile 128x64), 96 KB
    {"256x128x32s4", 256, 128, 256},   // 7: 4-deep, 96 KB
    {"256x256x32s3w16", 256, 256, 256}, // 8: 16 waves (4x4, wave tile 64x64), 96 KB ring, 1 block/CU
    {"256x256x64s2w16", 256, 256, 256}, // 9: 16 waves, 128 KB ring
    {"256x256x32s4w16", 256, 256, 256}, // 10: 16 waves, 4-deep 128 KB ring
    {"256x256x64s2w8", 256, 256, 256},  // 11: 8 waves (2x4, wave tile 128x64), 128 KB ring
    {"256x256x32s4w8", 256, 256, 256},  // 12: 8 waves, 4-deep ring of 32 KB stages
    {"256x256x64r2w16", 256, 256, 256}, // 13: as 9 but register-staged tiles (global_load -> ds_write)
    {"128x128x64r2", 128, 128, 512},    // 14: as 0 but register-staged
    {"256x128x64r2", 256, 128, 256},    // 15: as 3 but register-staged
    {"128x320x64s2", 128, 320, 256},    // 16: 8 waves (4x2, wave tile 32x160): N = 1280 / 640 outputs in exactly 4 / 2 tile columns
    {"256x256x64pp", 256, 256, 256},    // 17: 8 waves in two ping-pong groups, 16x16x32 MFMA, 4 phases per K tile (gemm_pp.hpp)
    {"128x320x64pp", 128, 320, 256},    // 18: the same structure on the 128x320 tile (wave 64x80)
    {"256x256x64ppsk", 256, 256, 256},  // 19: variant 17 as STREAM-K: the launch's K-tile units dealt evenly over the 256 blocks, partial tiles combined in-launch
    {"128x320x64pp2", 128, 320, 256},   // 20: variant 18 with TWO phases per K tile (20 MFMAs per barrier-to-barrier section instead of 8 / 12)
    {"256x256x64pp2", 256, 256, 256},   // 21: variant 17 with two phases per K tile (32 MFMAs per section)
    {"128x320x64ppb", 128, 320, 256},   // 22: variant 18 on the re-balanced staging schedule (gemm_pp.hpp SCH = 1)
    {"64x128x64s3", 64, 128, 512},      // 23: variant 1 with a 3-deep ring (72 KB, 2 blocks/CU): two K tiles in flight per block for grids that leave CUs half empty
    {"64x128x64r2", 64, 128, 768},      // 24: variant 1 register-staged (global_load -> ds_write): small grids are bound by the per-CU LDS-DMA issue rate
    {"256x128x64pp2", 256, 128, 256},   // 25: ping-pong tile for NARROW outputs (N = 128: the VAE's full-resolution convolutions), wave 128 x 32, two phases per K tile (16 MFMAs per section)
    {"256x256x64w4", 256, 256, 256},    // 26: ONE wave per SIMD (4 waves of 128 x 128, accumulators in AGPRs, software-pipelined inside the wave): gemm_w4.hip
    {"128x320x64w4", 128, 320, 256},    // 27: the same on the 128 x 320 tile (4 waves of 64 x 160)
    {"128x320x64ppsk", 128, 320, 256},  // 28: variant 18 as STREAM-K (round 4): N = 320 / 640 / 1280 outputs with FEW tiles and long K -- the 3x3 convolutions of SD1.5 batch 1
                                        //     (8192x320x2880: 64 tiles, 2048x640x5760: 32, 512x1280x11520: 16) -- dealt over all 256 CUs in K-tile units
    {"skinny128x64", 128, 64, 512},     // 29: M <= 128 weight streaming (gemm_skinny.hpp): all rows in one block, weights global -> registers, 7 K steps in flight, K slices + fixed-order reduce
    {"128x160x64tt", 128, 160, 512},    // 30: TWO tiles in flight per CU (gemm_tt.hip, round 5): 4-wave blocks, two resident per CU in two priority classes, so that one tile's residual
                                        //     read / output burst runs under the other tile's K loop -- the single-round 8192 x 1280 outputs of the SDXL transformer blocks
    {"conv3x3n16", 64, 16, 512},        // 31: 3x3 convolutions with Cout <= 16 over full-resolution maps (the VAE / TAESD output layers) as a STREAMING op (conv_smalln.hip, round 6): every wave walks
                                        //     a 16-pixel strip with a ring of input rows in its own LDS slice, weights in registers, one pass over the activations
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int pick_variant(const mlsd_gemm_args* a)
{
    if (g_gemm_variant >= 0 && g_gemm_variant < kNumVariants) return g_gemm_variant;
    if (a->xa_k && xattn_eligible(a) && pp_eligible(a, 128, 320)) return 18;      // the one tile that ends with the attention
    if (mlsd_conv_smalln_eligible(a)) return 31;      // a SHAPE rule above the table (A/B: MLSD_CONV_SMALLN=0): no GEMM tile is the right tool for Cout = 3
    if (a->tile_variant > 0 && a->tile_variant <= kNumVariants) return a->tile_variant - 1;
    if (a->M <= 64) return 1;
    // 128x128 (2 blocks/CU) vs 256x128 (1 block/CU): the bigger tile moves 25 % fewer operand bytes per
    // FLOP (measured +8..15 % on wide outputs) but quantises worse.  Pick by the fill of the last round
    // of blocks over the 256 CUs.
    auto fill = [&](int bm, int bn, int slots) {
        const long blocks = (long)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn);
        const long rounds = (blocks + slots - 1) / slots;
        return (double)blocks / (double)(rounds * slots);
    };
    const double e0 = fill(128, 128, 512), e3 = 1.08 * fill(256, 128, 256);
    if (a->M >= 2048 && a->N >= 1920 && e3 > e0) return 3;
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
    if (!a->C32 && !a->C16 && !a->xa_k) return mlsd_set_error(-1, "mlsd_gemm: no output");
    if (a->colstats && a->colstats_rows > 0 && mlsd_gemm_colstats_rows(a) != a->colstats_rows)
        return mlsd_set_error(-1, "mlsd_gemm: a GroupNorm was planned on this launch's column statistics (blocks of %d rows) but the launch "
                              "would write %d-row blocks / none: tile or epilogue settings changed after planning", a->colstats_rows, mlsd_gemm_colstats_rows(a));
    if (a->gn_y16 && !mlsd_gemm_gn_fused(a))
        return mlsd_set_error(-1, "mlsd_gemm: the plan dropped a GroupNorm for this launch's reduce pass but the launch would not run it (tile or K-split settings changed after planning)");
    if (a->xa_k && !mlsd_gemm_xattn_fused(a))
        return mlsd_set_error(-1, "mlsd_gemm: the plan dropped a cross attention for this projection's launch but the launch would not run it (tile settings changed after planning)");
    if (a->ln_y16 && !mlsd_gemm_ln_fused(a))
        return mlsd_set_error(-1, "mlsd_gemm: the plan dropped a LayerNorm for this launch's *_LN epilogue but the launch would not run it (tile or epilogue settings changed after planning)");
    hipStream_t st = (hipStream_t)stream;
    switch (pick_variant(a)) {
    case 1: return launch<64, 128, 64, 2, 2, 2>(a, st);
    case 3: return launch<256, 128, 64, 4, 2, 2>(a, st);
    case 4: return launch<256, 128, 32, 4, 2, 3>(a, st);
    case 9: return launch<256, 256, 64, 4, 4, 2>(a, st);
    case 27:
        if (mlsd_gemm_w4_eligible(a, 1)) return mlsd_gemm_w4(a, 1, st, g_gemm_ncu);
        if (pp_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true>(a, st);      // anything else: the ping-pong tile of the same shape
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: the 128x320 tiles do not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
    case 26:
        if (mlsd_gemm_w4_eligible(a, 0)) return mlsd_gemm_w4(a, 0, st, g_gemm_ncu);
        [[fallthrough]];                                                    // anything else: the ping-pong tile of the same shape
    case 17: return pp_eligible(a, 256, 256) ? launch_pp<256, 256, 2, 2, false>(a, st) : launch<256, 256, 64, 4, 4, 2>(a, st);
    case 19:
        if (sk_eligible(a, 256, 256)) return launch_pp<256, 256, 2, 2, false, true>(a, st);
        return pp_eligible(a, 256, 256) ? launch_pp<256, 256, 2, 2, false>(a, st) : launch<256, 256, 64, 4, 4, 2>(a, st);
    case 29:
        if (skinny_eligible(a)) return launch_skinny(a, st);
        return launch<64, 128, 64, 2, 2, 2>(a, st);
    case 30:
        if (mlsd_gemm_tt_eligible(a, tt_ncu())) return mlsd_gemm_tt(a, st, tt_ncu());
        if (pp_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true>(a, st);      // anything else: the ping-pong tile nearest in shape
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: the 128x320 tiles do not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
    case 31:
        if (mlsd_conv_smalln_eligible(a)) return mlsd_conv_smalln(a, st);
        return launch<128, 128, 64, 2, 2, 2>(a, st);
    case 28:
        if (sk_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true, true>(a, st);
        [[fallthrough]];
    case 18:
        if (pp_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true>(a, st);
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: the 128x320 tiles do not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
    case 20:
        if (pp_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true, false, 2>(a, st);
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: the 128x320 tiles do not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
#ifdef MLSD_GEMM_EXPERIMENTS   /* measured and not adopted (profiles/NOTES.md): the re-balanced staging schedule, the narrow 256x128 ping-pong tile */
    case 22:
        if (pp_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true, false, 4, 1>(a, st);
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: the 128x320 tiles do not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
    case 25: return pp_eligible(a, 256, 128) ? launch_pp<256, 128, 1, 1, false, false, 2>(a, st) : launch<256, 128, 64, 4, 2, 2>(a, st);
#else
    case 22:
        if (pp_eligible(a, 128, 320)) return launch_pp<128, 320, 3, 2, true>(a, st);
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: the 128x320 tiles do not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
    case 25: return launch<256, 128, 64, 4, 2, 2>(a, st);
#endif
    case 21: return pp_eligible(a, 256, 256) ? launch_pp<256, 256, 2, 2, false, false, 2>(a, st) : launch<256, 256, 64, 4, 4, 2>(a, st);
    case 16:
        if (a->act == MLSD_ACT_GEGLU) return mlsd_set_error(-1, "mlsd_gemm: tile variant 16 (odd slab count) does not support GEGLU");
        return launch<128, 320, 64, 4, 2, 2>(a, st);
#ifdef MLSD_GEMM_EXPERIMENTS   /* variants that lost the tile study on MI355X (kept reproducible, not built by default) */
    case 2: return launch<128, 128, 32, 2, 2, 4>(a, st);
    case 5: return launch<128, 128, 64, 2, 2, 3>(a, st);
    case 14: return launch<128, 128, 64, 2, 2, 2, true>(a, st);
    case 23: return launch<64, 128, 64, 2, 2, 3>(a, st);      /* round 3 small-M study (profiles/r3_gemm_smallm_rings.txt): no gain */
    case 24: return launch<64, 128, 64, 2, 2, 2, true>(a, st);
    case 6: return launch<256, 256, 32, 2, 4, 3>(a, st);
    case 7: return launch<256, 128, 32, 4, 2, 4>(a, st);
    case 8: return launch<256, 256, 32, 4, 4, 3>(a, st);
    case 10: return launch<256, 256, 32, 4, 4, 4>(a, st);
    case 11: return launch<256, 256, 64, 2, 4, 2>(a, st);
    case 12: return launch<256, 256, 32, 2, 4, 4>(a, st);
    case 13: return launch<256, 256, 64, 4, 4, 2, true>(a, st);
    case 15: return launch<256, 128, 64, 4, 2, 2, true>(a, st);
#endif
    default: return launch<128, 128, 64, 2, 2, 2>(a, st);
    }
}

MLSD_API void mlsd_gemm_set_panel(int width) { g_gemm_panel = width < 0 ? 0 : width; }   /* tile-order panel width (A/B timing) */
MLSD_API void mlsd_gemm_set_mode(int mode) { mlsd_gemm_set_panel(mode); }
MLSD_API void mlsd_gemm_force_variant(int v) { g_gemm_variant = v; }
MLSD_API void mlsd_gemm_set_epilogue(int e) { g_gemm_epi = e; }
MLSD_API void mlsd_gemm_set_debug(int d) { g_gemm_dbg = d; }
/* Round-5 experiment (profiles/r5_conv_korder_experiment.txt, tools/conv_korder_bench.py; TIMING ONLY -- the weights keep their (kh, kw, cin) order): the ping-pong convolutions
 * walk K slab by slab ((cin / 64, kh, kw, 64): the 9 taps of a 64-channel slab in consecutive K tiles) so that an input pixel's slab is re-read within 9 K tiles instead of once per
 * tap pass over all channels.  Measured 0 .. 4 % on the SDXL conv shapes once the clocks have settled: the long-K conv loops are not bound by that re-fetch (it hits the MALL), so
 * the weight re-layout + the three kernels' gather changes were not made.  EXPERIMENTS builds only. */
#ifdef MLSD_GEMM_EXPERIMENTS
MLSD_API void mlsd_gemm_set_korder(int k) { g_gemm_korder = k; }
#else
MLSD_API void mlsd_gemm_set_korder(int k) { (void)k; }
#endif
#ifdef MLSD_GEMM_EXPERIMENTS
MLSD_API void mlsd_gemm_set_splitk_inline(int on) { g_gemm_sk_inline = on != 0; }
#else
MLSD_API void mlsd_gemm_set_splitk_inline(int on) { (void)on; }      /* measured slower (profiles/NOTES.md): the code path is not in the product build */
#endif
/* 1 if the library was built with -DMLSD_GEMM_EXPERIMENTS (variants that lost their study: one-wave tiles, narrow / re-balanced ping-pong tiles, split-K reduced in
 * the launch, ping-pong attention); their tests skip otherwise */
MLSD_API int mlsd_has_experiments(void)
{
#ifdef MLSD_GEMM_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
MLSD_API void mlsd_gemm_set_splitk_parallel(int on) { g_gemm_sk_par = on ? 1 : 0; }
/* 1 if this split-K launch would add its K slices inside the launch (all blocks of a tile wait for each other: an in-launch hand-off the plan has to check) */
MLSD_API int mlsd_gemm_splitk_parallel(const mlsd_gemm_args* a)
{
    if (!a || a->ksplit < 2 || !a->sk_flags) return 0;
    const int v = pick_variant(a);
    if (v != 0 && v != 1) return 0;
    const int BM = v == 1 ? 64 : 128;
    const int nout = a->N;
    const bool vec = !(nout & 3) && (!a->C32 || (!(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15))) && (!a->C16 || (!(a->ldc16 & 3) && !((uintptr_t)a->C16 & 7))) &&
                     (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) && (!a->bias || !((uintptr_t)a->bias & 15)) &&
                     (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15))) && g_gemm_epi != 1;
    const int ns = vec ? splitk_slices(a, 64, nullptr) : 1;
    return ns > 1 && splitk_par_ok(a, BM, ns, (long)((a->M + BM - 1) / BM) * ((a->N + 127) / 128)) ? 1 : 0;
}
MLSD_API void mlsd_gemm_set_cus(int n) { g_gemm_ncu = n > 0 && n <= 256 ? n : 256; }
/* diagnostics: device buffer of 8 x uint64 per block (256 blocks) that the ping-pong kernels fill with s_memtime stamps:
 * [0] kernel entry, [1] prologue done, [2] first epilogue begins, [3] first epilogue issued, [4] last epilogue begins,
 * [5] last epilogue issued, [6] block exit (after the final wait), [7] tiles walked; NULL switches it off */
MLSD_API void mlsd_gemm_set_trace(void* buf) { g_gemm_tbuf = (unsigned long long*)buf; }

MLSD_API int mlsd_gemm_num_variants(void) { return kNumVariants; }

MLSD_API size_t mlsd_gemm_splitk_ws_bytes(int M, int N, int ksplit)
{   /* whole 128 x 128 tiles: the in-launch reduction keeps one slab per (tile, slice) */
    return ksplit > 1 ? (size_t)ksplit * ((M + 127) / 128 * 128) * ((N + 127) / 128 * 128) * sizeof(float) : 0;
}

MLSD_API size_t mlsd_gemm_streamk_ws_bytes(void) { return (size_t)256 * 256 * 256 * sizeof(float); }

MLSD_API int mlsd_gemm_colstats_rows(const mlsd_gemm_args* a)
{
    if (!a || !a->colstats) return 0;
    const int v = pick_variant(a);
    int bm, bn;
    /* a stream-K convolution has no statistics epilogue: handing it a GroupNorm's statistics would silently turn it into the plain 256 x 256 launch (round 3 shipped
     * exactly that: 24 tiles on 256 CUs, SD1.5's 2048x640x17280 at 366 us).  It keeps its tile; the GroupNorm keeps its first pass. */
    if (v == 19 && a->conv && sk_eligible(a, 256, 256, true)) return 0;
    if (v == 28 && a->conv && sk_eligible(a, 128, 320, true)) return 0;
    if ((v == 17 || v == 19 || v == 21 || v == 26) && pp_eligible(a, 256, 256)) { bm = 256; bn = 256; }   // (26 never takes a launch that has colstats set: it runs as 17)
    else if ((v == 18 || v == 20 || v == 22 || v == 27 || v == 28) && pp_eligible(a, 128, 320)) { bm = 128; bn = 320; }
#ifdef MLSD_GEMM_EXPERIMENTS
    else if (v == 25 && pp_eligible(a, 256, 128)) { bm = 256; bn = 128; }
#endif
    else {      // the general tiles that mlsd_gemm launches directly (round 4): {variant -> wave rows, BK}; every one of them has 64-column wave slabs
        switch (v) {
        case 0: return general_stats_rows(a, 64, 2, 64);       // 128x128x64, 2x2 waves
        case 1: return general_stats_rows(a, 32, 2, 64);       // 64x128x64, 2x2
        case 3: return general_stats_rows(a, 64, 2, 64);       // 256x128x64, 4x2
        case 4: return general_stats_rows(a, 64, 2, 32);       // 256x128x32s3, 4x2
        case 9: return general_stats_rows(a, 64, 2, 64);       // 256x256x64, 4x4
        default: return 0;
        }
    }
    const int e = pp_epilogue_kind(a, bn);
    return (e == PP_EPI_F32_STATS || e == PP_EPI_F32_RES_STATS) ? bm / 2 : 0;
}

/* 1 if this launch (with its ln_* fields set) would end with the LayerNorm of its output: the plan builder then drops the LayerNorm launch */
MLSD_API int mlsd_gemm_ln_fused(const mlsd_gemm_args* a)
{
    if (!a || !a->ln_y16) return 0;
    const int v = pick_variant(a);
    // 2: a split-K launch on the general tiles whose REDUCE pass ends with the LayerNorm (splitk_reduce_ln: one block per row; no in-launch hand-off)
    if ((v == 0 || v == 1) && a->ksplit > 1 && a->ws && a->C32 && !a->C16 && a->ln_gamma && a->ln_beta && !(a->N & 3) && a->N <= 4096 && !(a->ldln & 3) &&
        !((uintptr_t)a->ln_y16 & 7) && !((uintptr_t)a->ln_gamma & 15) && !((uintptr_t)a->ln_beta & 15) && a->act != MLSD_ACT_GEGLU && !g_gemm_sk_inline) {
        const bool vec = !(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15) && (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) &&
                         (!a->bias || !((uintptr_t)a->bias & 15)) && (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15))) && g_gemm_epi != 1;
        if (vec && splitk_slices(a, 64, nullptr) > 1 && !splitk_par_ok(a, v == 1 ? 64 : 128, splitk_slices(a, 64, nullptr), (long)((a->M + (v == 1 ? 63 : 127)) / (v == 1 ? 64 : 128)) * ((a->N + 127) / 128)))
            return 2;
    }
    if (v == 30) { const int e = mlsd_gemm_tt_eligible(a, tt_ncu()); return (e >= 4 && e <= 7) ? 1 : 0; }      // (TT_F32_LN / TT_F32_RES_LN / TT_CHAIN_*)
    if (v != 18 || !ln_eligible(a)) return 0;
    const int e = pp_epilogue_kind(a, 320);
    return (e == PP_EPI_F32_LN || e == PP_EPI_F32_RES_LN) ? 1 : 0;
}

/* 1 if this launch (gn_* fields set) ends its split-K reduce pass with the GroupNorm of its output: the plan builder then drops the GroupNorm launch */
MLSD_API int mlsd_gemm_gn_fused(const mlsd_gemm_args* a)
{
#ifndef MLSD_GEMM_EXPERIMENTS
    /* measured SLOWER than splitk_reduce + the one-dispatch GroupNorm on every map size of the SD1.5 b1 plan (evaluation 6.84 -> 6.88 / 6.90 ms, profiles/NOTES.md): the
     * reduce then runs on 64 blocks (one per image and group) instead of 640.  Only in EXPERIMENTS builds. */
    (void)a;
    return 0;
#else
    if (!a || !a->gn_y16 || a->ln_y16 || a->colstats) return 0;
    const int v = pick_variant(a);
    if ((v != 0 && v != 1) || a->ksplit < 2 || !a->ws || !a->C32 || a->C16 || !a->gn_gamma || !a->gn_beta || a->act == MLSD_ACT_GEGLU || g_gemm_sk_inline) return 0;
    if (a->gn_groups <= 0 || a->gn_hw <= 0 || (a->N % a->gn_groups) || (a->M % a->gn_hw)) return 0;
    const int cg = a->N / a->gn_groups;
    if ((cg & 3) || (long)a->gn_hw * (cg >> 2) > 256L * GNR_MAXI || (a->gn_ldy & 3) || ((uintptr_t)a->gn_y16 & 7) || ((uintptr_t)a->gn_gamma & 15) || ((uintptr_t)a->gn_beta & 15)) return 0;
    const bool vec = !(a->N & 3) && !(a->ldc32 & 3) && !((uintptr_t)a->C32 & 15) && (!a->resid || (!(a->ldr & 3) && !((uintptr_t)a->resid & 15))) &&
                     (!a->bias || !((uintptr_t)a->bias & 15)) && (!a->rowbias || (!(a->ldrb & 3) && !((uintptr_t)a->rowbias & 15))) && g_gemm_epi != 1;
    if (!vec || splitk_slices(a, 64, nullptr) < 2) return 0;
    return splitk_par_ok(a, v == 1 ? 64 : 128, splitk_slices(a, 64, nullptr), (long)((a->M + (v == 1 ? 63 : 127)) / (v == 1 ? 64 : 128)) * ((a->N + 127) / 128)) ? 0 : 1;
#endif
}

MLSD_API void mlsd_gemm_set_xattn(int mode) { g_xattn_mode = (mode >= 0 && mode <= 2) ? mode : -1; }

/* 1 if this launch (xa_* fields set) ends with the cross attention of the q it projects: the plan builder then records no attention launch */
MLSD_API int mlsd_gemm_xattn_fused(const mlsd_gemm_args* a)
{
    if (!a || !a->xa_k || g_gemm_variant >= 0) return 0;
    return (xattn_eligible(a) && pp_eligible(a, 128, 320) && pick_variant(a) == 18) ? 1 : 0;
}

MLSD_API const char* mlsd_gemm_variant(const mlsd_gemm_args* a)
{
    static thread_local char buf[64];
    int v = pick_variant(a);
    if (v == 19 && !sk_eligible(a, 256, 256)) v = 17;
    if (v == 28 && !sk_eligible(a, 128, 320)) v = 18;
    if (v == 29 && !skinny_eligible(a)) v = 1;
    if (v == 29) {
        snprintf(buf, sizeof(buf), "gemm<%s,%s,k/%d>", kVariants[v].name, a->conv ? "conv" : "linear", skinny_slices(a, nullptr));
        return buf;
    }
    if (v == 30 && !mlsd_gemm_tt_eligible(a, tt_ncu())) v = 18;
    if (v == 31 && !mlsd_conv_smalln_eligible(a)) v = 0;
    if (v == 26 && !mlsd_gemm_w4_eligible(a, 0)) v = 17;
    if (v == 27 && !mlsd_gemm_w4_eligible(a, 1)) v = 18;
    if ((v == 17 || v == 21) && !pp_eligible(a, 256, 256)) v = 9;
    if ((v == 18 || v == 20 || v == 22) && !pp_eligible(a, 128, 320)) v = 16;
#ifdef MLSD_GEMM_EXPERIMENTS
    if (v == 25 && !pp_eligible(a, 256, 128)) v = 3;
#else
    if (v == 25) v = 3;
    if (v == 22) v = 18;
#endif
    const int bk = strstr(kVariants[v].name, "x32s") ? 32 : 64;
    const int ns = ((v >= 17 && v <= 22) || v >= 25) ? 1 : splitk_slices(a, bk, nullptr);   /* (the persistent tiles never split K over the grid) */
    if (ns > 1) snprintf(buf, sizeof(buf), "gemm<%s,%s%s,k/%d%s>", kVariants[v].name, a->conv ? "conv" : "linear", mlsd_gemm_ln_fused(a) == 2 ? "+layernorm" : (mlsd_gemm_gn_fused(a) ? "+groupnorm" : ""), ns, mlsd_gemm_splitk_parallel(a) ? "p" : "");
    else if (mlsd_gemm_xattn_fused(a)) snprintf(buf, sizeof(buf), "gemm<%s,linear+attention>", kVariants[v].name);      /* the q projection of a cross attention that ends with it */
    else if (mlsd_gemm_ln_fused(a)) snprintf(buf, sizeof(buf), "gemm<%s,linear+layernorm>", kVariants[v].name);      /* the launch ends with the LayerNorm of its output */
    else snprintf(buf, sizeof(buf), "gemm<%s,%s>", kVariants[v].name, a->conv ? "conv" : "linear");
    return buf;
}

}  // extern "C"
