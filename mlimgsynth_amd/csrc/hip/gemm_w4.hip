// gemm_w4: 256 x 256 x 64 GEMM tile with ONE wave per SIMD (round 3).
//
// Why a third tile family.  The ping-pong tiles (gemm_pp.hpp) put two waves on every SIMD; each wave owns 128 x 64 (or 64 x 80) outputs,
// so per K tile the 8 waves read (128 + 64) x 128 B = 24 KB of fragments each: 192 KB of ds_read traffic + 64 KB of LDS-DMA writes =
// 256 KB through the CU's 128 B/clock LDS port = 2048 clocks, exactly the 2048 clocks the K tile's 128 MFMAs per SIMD take.  The port
// is as busy as the matrix pipes, and the measured loop (2625 clocks per K tile, profiles/r3_gemm_loop_ablation.txt) shows it; the
// operand fill itself is not the limit (tools/fill_probe.py: 46 B/clock/CU of LDS-DMA out of L2, the loop needs 31).  Register
// blocking is the lever: FOUR waves of 128 x 128 read (128 + 128) x 128 B = 32 KB each = 128 KB per K tile (+ 64 KB of DMA = 1536
// clocks of the port against 2048 of matrix work).  A wave then holds 256 accumulator registers (the unified 512-register file of a
// SIMD that runs a single wave: accumulators in the AGPR half) and there is no second wave to hide its fragment reads behind, so the
// loop is software-pipelined inside the wave instead:
//
//   K tile t, ring stage s = t & 1, two halves (k-steps of 32):
//     H1: 64 MFMAs on the ks = 0 fragments | ds_reads of the ks = 1 fragments of tile t
//         s_waitcnt vmcnt(0) (the LDS-DMA of tile t+1, issued one tile ago) ; s_barrier
//     H2: 64 MFMAs on the ks = 1 fragments | ds_reads of the ks = 0 fragments of tile t+1 | LDS-DMA of tile t+2 into stage s
//   One barrier per K tile, in the middle: after it every wave has read both k-steps of tile t (stage s is free for tile t+2) and
//   every wave's share of tile t+1 has landed (its ks = 0 fragments may be read).  Fragment reads and staging instructions are
//   interleaved with the MFMAs by sched_group_barrier groups (1 LDS / VMEM instruction per 4 MFMAs).
//
// Layout, swizzle, fragment map and the transposed product (a lane holds 4 consecutive output columns, epilogue straight from the
// accumulators, fp16 rows as 16-byte stores after v_permlane16_swap) are those of gemm_pp.hpp.  Persistent: grid = min(tiles, 256),
// block b walks tiles b, b + G, ...; the first two K tiles of the next output tile are put in flight before the epilogue of the
// current one.  Linear problems only (the UNet's fused projections), M, N multiples of 128, K a multiple of 64 and >= 128; epilogues:
// fp16, GEGLU -> fp16, fp32, fp32 + fp32 residual (bias optional).  Everything else stays on the ping-pong tiles.
#include "common.hpp"
#include "mlsd_kernels.h"
#include <type_traits>

namespace {

struct W4P {
    const _Float16 *A, *B;
    long lda, ldb;
    int M, N, K;
    const float* bias;
    const float* resid; long ldr;
    float* C32; long ldc32;
    _Float16* C16; long ldc16;
    int nbm, nbn, gw;
};

enum { W4_F16 = 1, W4_F32 = 2, W4_F32_RES = 3, W4_GEGLU16 = 4 };

template <int N>
__device__ __forceinline__ void w4_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XCD-aware bijective remap of the virtual block index + column panels (as gemm_pp.hpp pp_tile_coords)
template <int BM, int BN>
__device__ __forceinline__ void w4_tile_coords(const W4P& p, int v, int& m0, int& n0)
{
    const int nblk = p.nbm * p.nbn;
    const int q = nblk >> 3, r = nblk & 7, x = v & 7, j = v >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    int bm, bn;
    if (p.gw > 0 && p.nbn > p.gw) {
        const int per_panel = p.gw * p.nbm;
        const int panel = bid / per_panel;
        const int first = panel * p.gw;
        const int w = min(p.gw, p.nbn - first);
        const int rr = bid - panel * per_panel;
        bm = rr / w; bn = first + (rr - bm * w);
    } else {
        bm = bid / p.nbn; bn = bid - bm * p.nbn;
    }
    m0 = bm * BM; n0 = bn * BN;
}

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const W4P p)
{
    constexpr int BK = 64, RB = BK * 2;
    constexpr int A_ST = BM * RB, B_ST = BN * RB, BBASE = 2 * A_ST;   // LDS: two A stages, then THREE B stages (the weights come cold from HBM
                                                                      // in the plan: their staging runs one K tile further ahead than the activations')
    constexpr int WM = BM / 2, WN = BN / 2;        // waves 2 x 2
    constexpr int NI = WM / 16, NC = WN / 16;      // 16-row / 16-column blocks of the wave's tile (256 x 256: 8 x 8; 128 x 320: 4 x 10)
    constexpr int NS = NI + NC;                    // fragment reads per k-step = staging instructions per K tile = slices of a half
    static_assert(BM % 64 == 0 && BN % 64 == 0 && NI * NC * 4 <= 256, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nkt = p.K / BK;
    const int nblk = p.nbm * p.nbn, G = gridDim.x;

    // staging: wave-instruction i (0..7) of wave w fills tile rows 8 (4 i + w) .. + 7 (lane l -> row l >> 3, physical chunk l & 7); the
    // logical chunk whose swizzled slot that is:  (l & 7) ^ ((row >> 1) & 7), and (row >> 1) & 7 = 4 (w & 1) + (l >> 4) for every i
    const int srow = wave * 8 + (lane >> 3);                                   // + 32 i
    const int schunk = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;   // halfs
    const _Float16 *pa[NI], *pb[NC];               // (BM / 32 = NI row groups of A, BN / 32 = NC of B per wave)
    auto enter = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; ++i) pa[i] = p.A + (long)min(m0 + srow + 32 * i, p.M - 1) * p.lda + schunk;
#pragma unroll
        for (int i = 0; i < NC; ++i) pb[i] = p.B + (long)min(n0 + srow + 32 * i, p.N - 1) * p.ldb + schunk;
    };
    // staging instruction g (0..NI-1: A row groups, NI..NS-1: B row groups) of this wave for the K tile the running pointers stand at
    auto issue_one = [&](int g, int sa, int sb) __attribute__((always_inline)) {
        if (g < NI) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[g],
                                             (__attribute__((address_space(3))) void*)(smem + sa * A_ST + wave * 1024 + g * 4096), 16, 0, 0);
            pa[g] += BK;
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[g - NI],
                                             (__attribute__((address_space(3))) void*)(smem + BBASE + sb * B_ST + wave * 1024 + (g - NI) * 4096), 16, 0, 0);
            pb[g - NI] += BK;
        }
    };
    auto issue_A = [&](int sa) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NI; ++g) issue_one(g, sa, 0);
    };
    auto issue_B = [&](int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int g = NI; g < NS; ++g) issue_one(g, 0, sb);
    };

    // fragments (16x16x32: lane = (row l15, k group lg)); the swizzle term is lane-constant
    const int c0 = lg ^ (l15 >> 1);
    const int fk[2] = {c0 << 4, (c0 ^ 4) << 4};
    const int fa = (wr * WM + l15) * RB, fb = BBASE + (wc * WN + l15) * RB;
    f16x8 af[2][NI], bf[2][NC];
    auto read_one = [&](int ks, int g, int sa, int sb) __attribute__((always_inline)) {     // g 0..NI-1: A block g, NI..NS-1: B block g - NI
        if (g < NI) af[ks][g] = *reinterpret_cast<const f16x8*>(smem + sa * A_ST + fk[ks] + fa + g * 16 * RB);
        else bf[ks][g - NI] = *reinterpret_cast<const f16x8*>(smem + sb * B_ST + fk[ks] + fb + (g - NI) * 16 * RB);
    };
    f32x4 acc[NI][NC];
    // one half of a K tile, hand-interleaved in NS slices: 1 fragment read of the NEXT half + (ISSUE) 1 staging instruction + NI NC / NS MFMAs of
    // this half (256 x 256: 16 slices of 4; 128 x 320: 14 slices of 2-3).  sched_barrier keeps the slices as written (left to itself the scheduler put the 16 reads and 16 staging instructions in one
    // clump behind 4 MFMAs: the M0 set-up of every LDS-DMA instruction defeats its sched_group_barrier pipelines).
    auto half = [&](int ks, int ks_rd, int sa_rd, int sb_rd, auto ISSUE_A_, auto ISSUE_B_, int sa_wr, int sb_wr) __attribute__((always_inline)) {
        constexpr bool ISSUE_A = decltype(ISSUE_A_)::value, ISSUE_B = decltype(ISSUE_B_)::value;
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            read_one(ks_rd, g, sa_rd, sb_rd);
            if constexpr (ISSUE_A) { if (g < NI) issue_one(g, sa_wr, sb_wr); }
            if constexpr (ISSUE_B) { if (g >= NI) issue_one(g, sa_wr, sb_wr); }
#pragma unroll
            for (int q = g * (NI * NC) / NS; q < (g + 1) * (NI * NC) / NS; ++q)
                acc[q / NC][q % NC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][q % NC], af[ks][q / NC], acc[q / NC][q % NC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // stores of one epilogue per lane (they are YOUNGER than the next tile's first two staging units, issued before the epilogue: the
    // counted waits at the head of the next tile leave them in flight; vmcnt holds 6 bits).
    // (Running the K tiles of consecutive output tiles as ONE stream, like the ping-pong kernels do -- the next tile's first units in the
    // staging slots of this tile's last two K tiles -- keeps the 2 NS fragment registers live across the epilogue: 56 (128 x 320) to 600
    // (256 x 256) spilled registers.  Hence a short prologue per output tile; the tiles this kernel is selected for are single-round.)
    constexpr int NST = EPI == W4_F16 ? NI * NC / 2 : EPI == W4_GEGLU16 ? NI * NC / 4 : NI * NC;
    constexpr int W_HEAD = NI + 2 * NC, W_HEAD_ST = W_HEAD + NST > 63 ? 63 : W_HEAD + NST;       // younger than A(0), B(0): A(1), B(1), B(2) [+ the stores]
    constexpr int W_MID = NC, W_MID_ST = NC + NST > 63 ? 63 : NC + NST;                          // younger than A(t+1), B(t+1): B(t+2) [+ the stores]
    bool stored = false;                           // this wave ran the previous tile's epilogue
    int sa = 0, sb = 0;                            // ring stages of the current K tile (A: t & 1, B: t % 3)
    // K tile t:  H1: MFMAs of k-step 0 | reads of k-step 1 of this tile;   counted wait + barrier: A(t+1), B(t+1) have landed and every wave is
    // done with this tile's stages;   H2: MFMAs of k-step 1 | reads of k-step 0 of tile t+1 (past the end: stale, unused) | staging of A(t+2) into
    // this tile's A stage and of B(t+3) into its B stage (A units first: the counted wait may leave exactly the NC youngest, B(t+2), in flight)
    auto body = [&](int t, auto ISSUE_A_, auto ISSUE_B_) __attribute__((always_inline)) {
        const int sb1 = sb == 2 ? 0 : sb + 1;
        half(0, 1, sa, sb, std::false_type{}, std::false_type{}, 0, 0);
        if (t == 0 && stored) w4_wait_vmcnt<W_MID_ST>();
        else if (t + 2 < nkt) w4_wait_vmcnt<W_MID>();
        else w4_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the k-step-1 fragments of this stage must have returned before the staging behind the barrier may overwrite it: gemm_tt.hip)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        half(1, 0, sa ^ 1, sb1, ISSUE_A_, ISSUE_B_, sa, sb);
        sa ^= 1; sb = sb1;
    };
    auto tile_head = [&]() __attribute__((always_inline)) {      // the first units of an output tile: A(0) B(0) A(1) B(1) B(2), in that order (K >= 192)
        issue_A(0); issue_B(0); issue_A(1); issue_B(1); issue_B(2);
    };

    int m0, n0;
    w4_tile_coords<BM, BN>(p, (int)blockIdx.x, m0, n0);
    enter(m0, n0);
    tile_head();
    for (int v = (int)blockIdx.x; v < nblk; v += G) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (stored) w4_wait_vmcnt<W_HEAD_ST>(); else w4_wait_vmcnt<W_HEAD>();      // K tile 0 of this output tile has landed
        __builtin_amdgcn_s_barrier();
        sa = 0; sb = 0;
#pragma unroll
        for (int g = 0; g < NS; ++g) read_one(0, g, 0, 0);
        int t = 0;
        for (; t + 3 < nkt; ++t) body(t, std::true_type{}, std::true_type{});
        for (; t + 2 < nkt; ++t) body(t, std::true_type{}, std::false_type{});
        for (; t < nkt; ++t) body(t, std::false_type{}, std::false_type{});
        stored = false;
        // ---- the next output tile's first units go in flight before this tile's epilogue
        const int wrow0 = m0 + wr * WM, wcol0 = n0 + wc * WN;
        if (v + G < nblk) {
            w4_tile_coords<BM, BN>(p, v + G, m0, n0);
            enter(m0, n0);
            tile_head();
        }
        if (wrow0 >= p.M || wcol0 >= p.N) continue;            // M, N multiples of 128: a wave's block is all in or all out
        stored = true;
        // ---- epilogue: acc[i][c][e] = C[wrow0 + 16 i + l15][wcol0 + 16 c + 4 lg + e]
        f32x4 cb[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) cb[c] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wcol0 + c * 16 + 4 * lg) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == W4_F16) {
            _Float16* rowp = p.C16 + (long)(wrow0 + l15) * p.ldc16 + wcol0 + 8 * (lg >> 1) + 16 * (lg & 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int c = 0; c < NC; c += 2) {
                    const f32x4 v0 = acc[i][c] + cb[c], v1 = acc[i][c + 1] + cb[c + 1];
                    const f16x4 h0 = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3]};
                    const f16x4 h1 = {(_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                    const u32x2 a = __builtin_bit_cast(u32x2, h0), b = __builtin_bit_cast(u32x2, h1);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    *reinterpret_cast<u32x4*>(rowp + (long)i * 16 * p.ldc16 + c * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
            }
        } else if constexpr (EPI == W4_GEGLU16 && WN % 64 == 0) {
            // weight rows interleaved in blocks of 32 (value | gate): column blocks 4g, 4g+1 = value, 4g+2, 4g+3 = gate of output columns (col >> 6) * 32 ..
            _Float16* rowp = p.C16 + (long)(wrow0 + l15) * p.ldc16 + (wcol0 >> 6) * 32 + 8 * (lg >> 1) + 16 * (lg & 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int g = 0; g < WN / 64; ++g) {
                    u32x2 hh[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 a4 = acc[i][4 * g + j] + cb[4 * g + j], g4 = acc[i][4 * g + 2 + j] + cb[4 * g + 2 + j];
                        const f16x4 h = {(_Float16)(a4[0] * gelu_tanh_f(g4[0])), (_Float16)(a4[1] * gelu_tanh_f(g4[1])),
                                         (_Float16)(a4[2] * gelu_tanh_f(g4[2])), (_Float16)(a4[3] * gelu_tanh_f(g4[3]))};
                        hh[j] = __builtin_bit_cast(u32x2, h);
                    }
                    const auto r0 = __builtin_amdgcn_permlane16_swap(hh[0][0], hh[1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(hh[0][1], hh[1][1], false, false);
                    *reinterpret_cast<u32x4*>(rowp + (long)i * 16 * p.ldc16 + g * 32) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
            }
        } else {
            float* rowp = p.C32 + (long)(wrow0 + l15) * p.ldc32 + wcol0 + 4 * lg;
            const float* resp = EPI == W4_F32_RES ? p.resid + (long)(wrow0 + l15) * p.ldr + wcol0 + 4 * lg : nullptr;
            // residual rows are fetched PF row blocks ahead of their use (all of them on the 128 x 320 tile: 160 registers the main loop's fragments
            // no longer need): one exposed HBM round trip per output tile instead of one per row block
            constexpr int PF = EPI == W4_F32_RES ? (NI * NC <= 40 ? NI : 2) : 0;
            f32x4 rr[EPI == W4_F32_RES ? NI : 1][EPI == W4_F32_RES ? NC : 1];
            auto fetch = [&](int i) __attribute__((always_inline)) {
                if constexpr (EPI == W4_F32_RES) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) rr[i][c] = *reinterpret_cast<const f32x4*>(resp + (long)i * 16 * p.ldr + c * 16);
                }
            };
#pragma unroll
            for (int i = 0; i < PF && i < NI; ++i) fetch(i);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (i + PF < NI) fetch(i + PF);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    f32x4 v4 = acc[i][c] + cb[c];
                    if constexpr (EPI == W4_F32_RES) v4 += rr[i][c];
                    *reinterpret_cast<f32x4*>(rowp + (long)i * 16 * p.ldc32 + c * 16) = v4;
                }
            }
        }
    }
}


// (The same loop on v_mfma_f32_32x32x16_f16 -- twice the FLOPs per instruction, half the operand-register reads per FLOP -- was built for one measurement
// and removed: 1094 - 1133 TFLOP/s on 8192^3 against 1330 - 1345 for the 16x16x32 form, bit-identical results; profiles/r3_gemm_w4.txt.)

int g_w4_panel = 8;

}  // namespace

// problems these tiles take (mlsd_gemm tile variants 26 / 27 hand everything else to the ping-pong tile of the same shape); tile 0: 256 x 256,
// 1: 128 x 320.  Returns the epilogue kind (W4_*) or 0.
extern "C" int mlsd_gemm_w4_eligible(const mlsd_gemm_args* a, int tile)
{
    const int wm = tile ? 64 : 128, wn = tile ? 160 : 128;
    if (a->conv || (a->K & 63) || a->K < 192 || (a->M % wm) || (a->N % wn)) return 0;
    if (a->rowbias || a->bias_m || a->colstats) return 0;
    const bool c16_wide = a->C16 && !(a->ldc16 & 7) && !((uintptr_t)a->C16 & 15);
    if (a->bias && ((uintptr_t)a->bias & 15)) return 0;
    if (a->act == MLSD_ACT_GEGLU) return (!tile && c16_wide && !a->C32 && !a->resid) ? W4_GEGLU16 : 0;
    if (a->act != MLSD_ACT_NONE) return 0;
    if (a->C16 && !a->C32 && !a->resid) return c16_wide ? W4_F16 : 0;
    if (a->C32 && !a->C16) {
        if ((a->ldc32 & 3) || ((uintptr_t)a->C32 & 15)) return 0;
        if (!a->resid) return W4_F32;
        return (!(a->ldr & 3) && !((uintptr_t)a->resid & 15)) ? W4_F32_RES : 0;
    }
    return 0;
}

template <int BM, int BN>
static int w4_launch(const mlsd_gemm_args* a, int epi, void* stream, int ncu)
{
    W4P p;
    p.A = (const _Float16*)a->A; p.B = (const _Float16*)a->W_; p.lda = a->lda; p.ldb = a->ldb; p.M = a->M; p.N = a->N; p.K = a->K;
    p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr; p.C32 = a->C32; p.ldc32 = a->ldc32; p.C16 = (_Float16*)a->C16; p.ldc16 = a->ldc16;
    p.nbm = (a->M + BM - 1) / BM; p.nbn = (a->N + BN - 1) / BN; p.gw = g_w4_panel;
    const int ntiles = p.nbm * p.nbn;
    constexpr size_t LDS = (2 * (size_t)BM + 3 * (size_t)BN) * 64 * 2;      // two A stages + three B stages
    const dim3 grid(ntiles < ncu ? ntiles : ncu), block(256);
    auto go = [&](auto kfn) -> int {
        MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        hipLaunchKernelGGL(kfn, grid, block, LDS, (hipStream_t)stream, p);
        return mlsd_check_launch("gemm_w4_kernel");
    };
    switch (epi) {
    case W4_F16: return go(gemm_w4_kernel<BM, BN, W4_F16>);
    case W4_GEGLU16:
        if constexpr (BN == 256) return go(gemm_w4_kernel<BM, BN, W4_GEGLU16>);
        return mlsd_set_error(-1, "mlsd_gemm_w4: GEGLU needs the 256-wide tile");
    case W4_F32: return go(gemm_w4_kernel<BM, BN, W4_F32>);
    default: return go(gemm_w4_kernel<BM, BN, W4_F32_RES>);
    }
}

extern "C" int mlsd_gemm_w4(const mlsd_gemm_args* a, int tile, void* stream, int ncu)
{
    const int epi = mlsd_gemm_w4_eligible(a, tile);
    if (!epi) return mlsd_set_error(-1, "mlsd_gemm_w4: problem not eligible for the one-wave-per-SIMD tiles");
    return tile ? w4_launch<128, 320>(a, epi, stream, ncu) : w4_launch<256, 256>(a, epi, stream, ncu);
}
