// gemm_tt: 128 x 160 x 64 GEMM tile on FOUR waves, TWO blocks resident per CU ("two tiles in flight"; round 5, tile variant 30).
//
// Why.  A launch of 128 x 320 ping-pong tiles gives every CU ONE block: all CUs loop, then all CUs read their residual tile and burst their output with the matrix pipes idle
// (profiles/NOTES.md "Round 5").  Two HALF tiles per CU in flight let one tile's burst run under the other tile's K loop -- if they are out of step.
// MEASURED (tools/gemm_tt_trace.py, 100 MHz stamps + HW_ID): they are, without help -- the block dispatched first on a CU gets the issue slots (oldest first): on 8192x1280x1280
// class 0 (blocks 0..255) ends its loop at 19.1 us and exits at 24.6 while class 1 loops until 27.4 and exits at 33.2; an explicit s_setprio for class 0 (kept: mlsd_gemm_tt_set_prio)
// changes nothing.  What the stagger buys is eaten on the full-chip single-round launches that end with a LayerNorm (the two co-resident loops run at 880 TFLOP/s against 1220 for
// the ping-pong loop: +0.4 .. +0.7 % in the plan), so the plan uses this kernel where it wins (same-box A/B, profiles/r5_gemm_tt_*.txt, r5_tune_inplan_tt_candidate.txt):
//   * launches whose 128 x 320 tiles would fill at most half of the CUs (SDXL batch 1 / 2: evaluation -1 .. -3.5 %),
//   * fp16-output and 1 x 1 launches, chosen by the in-plan tile pass (SDXL b4: the cross-attention q projections 36.2 -> 33.5 us, skip convolutions -10 .. -25 %),
//   * LayerNorm producers on general tiles with N <= 640 (SD1.5: 30 LayerNorm dispatches gone).
// The tiles of a row block are consecutive blocks of one XCD and share a class (they exchange LayerNorm statistics and must not wait for a slower partner).
//
// Block: 4 waves 2 x 2, wave tile 64 x 80 (the ping-pong kernel's: 80 accumulator registers), one wave per SIMD; the second block of the CU supplies the second wave per SIMD.
// LDS: two stages of A (128 rows) + two of B (160 rows) x 128 B = 72 KB (+ 2 KB for the *_LN epilogue): two blocks per CU.  The loop is software-pipelined inside the wave
// as in gemm_w4.hip (H1: MFMAs of k-step 0 | fragment reads of k-step 1; counted wait + ONE barrier per K tile; H2: MFMAs of k-step 1 | reads of the next tile's k-step 0 |
// LDS-DMA of the tile after that into the stage just released).  Layout, swizzle, fragment map, transposed product and the epilogue bodies are those of gemm_pp.hpp.
// Not persistent: one tile per block, grid = tiles; partner tiles (the N / 160 column tiles of a row block) are consecutive block numbers of one XCD.
// Linear problems only: M % 128 == 0, N % 160 == 0, K % 64 == 0, K >= 128; epilogues fp16 | fp32 | fp32 + residual | the latter two ending with the LayerNorm of the rows
// (mlsd_gemm_args.ln_*; the exchange of gemm_pp.hpp's *_LN epilogues with N / 160 partners).
#include "common.hpp"
#include "mlsd_kernels.h"
#include <type_traits>
#include <stdlib.h>

namespace {

struct TTP {
    const _Float16 *A, *B;
    long lda, ldb;
    int M, N, K;
    const float* bias;
    const float* resid; long ldr;
    float* C32; long ldc32;
    _Float16* C16; long ldc16;
    int nbm, nbn, prio;
    const float *ln_g, *ln_b; float ln_eps; _Float16* ln_y; long ldln; float* ln_ws; unsigned* ln_cnt; int ln_slot;
    unsigned long long* tbuf;      // diagnostics: 4 words per block {start, loop end, exit (100 MHz), HW_ID | XCC_ID << 32}
};

enum { TT_F16 = 1, TT_F32 = 2, TT_F32_RES = 3, TT_F32_LN = 4, TT_F32_RES_LN = 5 };

template <int N>
__device__ __forceinline__ void tt_wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tt_kernel(const TTP p)
{
    constexpr int BM = 128, BN = 160, BK = 64, RB = BK * 2;
    constexpr int A_ST = BM * RB, B_ST = BN * RB, BBASE = 2 * A_ST, RING = 2 * (A_ST + B_ST);
    constexpr int WM = BM / 2, WN = BN / 2;        // waves 2 x 2
    constexpr int NI = WM / 16, NC = WN / 16;      // 4 x 5 blocks of 16 x 16
    constexpr int NS = NI + NC;                    // fragment reads per k-step = staging instructions per K tile per wave (BM / 32 + BN / 32)
    static_assert(BM / 32 == NI && BN / 32 == NC, "staging instructions = fragment reads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nkt = p.K / BK;

    // tile of this block: XCD x = blockIdx & 7 walks a contiguous range of the row-major tile order (the nbn column tiles of a row block are neighbours there)
    const int nblk = p.nbm * p.nbn, v = (int)blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, x8 = v & 7, j8 = v >> 3;
    const int bid = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + j8;
    const int bmi = bid / p.nbn, bni = bid - bmi * p.nbn;
    const int m0 = bmi * BM, n0 = bni * BN;
    // priority class: blocks 0..255 (the first one on every CU), 512..767, ... run ahead of the others
    if (p.prio && !((v >> 8) & 1)) __builtin_amdgcn_s_setprio(2);
    // *_LN endings: the tag of the records this tile wrote in the PREVIOUS launch of this op (its own scratch region: mlblock.c wire_ln_fold; 0 = never) -- this launch's
    // records carry that + 1.  Every wave reads the tag of the first row IT publishes, long before it publishes: no other block or wave can have changed it, and the partner
    // tiles of the row block left the previous launch with the same tag.  (First form of round 6: one epoch word per launch, read at kernel entry and advanced by tile (0, 0)
    // once it had its partners' records -- a block that ENTERED after that took the next launch's tag and starved its row block: about one give-up in 20 000 launches of the
    // 512-block grids in tools/soak_r5.py.)
    unsigned ln_epoch = 0;
    if constexpr (EPI == TT_F32_LN || EPI == TT_F32_RES_LN)
        ln_epoch = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.ln_ws) + (((long)bmi * p.nbn + bni) * BM + wr * 64) * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.tbuf && tid == 0) {
        p.tbuf[(long)v * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        p.tbuf[(long)v * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }

    // staging: wave-instruction i of wave w fills tile rows 8 (4 i + w) .. + 7 (lane l -> row l >> 3, physical chunk l & 7)
    const int srow = wave * 8 + (lane >> 3);                                      // + 32 i
    const int schunk = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;   // halfs
    const _Float16* pa = p.A + (long)(m0 + srow) * p.lda + schunk;
    const _Float16* pb = p.B + (long)(n0 + srow) * p.ldb + schunk;
    long a32 = 32 * p.lda, b32 = 32 * p.ldb;
    auto issue_one = [&](int g, int st) __attribute__((always_inline)) {          // g 0..NI-1: A row groups, NI..NS-1: B row groups; K tile = where pa / pb stand
        if (g < NI) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + g * a32),
                                             (__attribute__((address_space(3))) void*)(smem + st * A_ST + wave * 1024 + g * 4096), 16, 0, 0);
        } else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + (g - NI) * b32),
                                             (__attribute__((address_space(3))) void*)(smem + BBASE + st * B_ST + wave * 1024 + (g - NI) * 4096), 16, 0, 0);
    };
    auto issue_tile = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NS; ++g) issue_one(g, st);
        pa += BK; pb += BK;
    };

    const int c0 = lg ^ (l15 >> 1);
    const int fk[2] = {c0 << 4, (c0 ^ 4) << 4};
    const int fa = (wr * WM + l15) * RB, fb = BBASE + (wc * WN + l15) * RB;
    f16x8 af[2][NI], bf[2][NC];
    auto read_one = [&](int ks, int g, int st) __attribute__((always_inline)) {
        if (g < NI) af[ks][g] = *reinterpret_cast<const f16x8*>(smem + st * A_ST + fk[ks] + fa + g * 16 * RB);
        else bf[ks][g - NI] = *reinterpret_cast<const f16x8*>(smem + st * B_ST + fk[ks] + fb + (g - NI) * 16 * RB);
    };
    f32x4 acc[NI][NC];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one half of a K tile in NS slices: 1 fragment read of the NEXT half + (ISSUE) 1 staging instruction + 2-3 MFMAs of this half
    auto half = [&](int ks, auto READ_, int ks_rd, int st_rd, auto ISSUE_, int st_wr) __attribute__((always_inline)) {
        constexpr bool READ = decltype(READ_)::value, ISSUE = decltype(ISSUE_)::value;
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            if constexpr (READ) read_one(ks_rd, g, st_rd);
            if constexpr (ISSUE) issue_one(g, st_wr);
#pragma unroll
            for (int q = g * (NI * NC) / NS; q < (g + 1) * (NI * NC) / NS; ++q)
                acc[q / NC][q % NC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][q % NC], af[ks][q / NC], acc[q / NC][q % NC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ISSUE) { pa += BK; pb += BK; }
    };
    // K tile t in stage s = t & 1.  H1: MFMAs of k-step 0 | reads of k-step 1;  wait (tile t + 1 has landed) + barrier (every wave has read both k-steps of tile t);
    // H2: MFMAs of k-step 1 | reads of k-step 0 of tile t + 1 | LDS-DMA of tile t + 2 into stage s
    auto body = [&](int s, auto NEXT_, auto ISSUE_) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(NEXT_)::value;
        half(0, std::true_type{}, 1, s, std::false_type{}, 0);
        if constexpr (NEXT) {
            // WAR on stage s: the k-step-1 fragments read above are consumed AFTER the barrier, and the first thing another wave does behind the barrier is LDS-DMA into
            // stage s.  Weight K tiles that the CU's other block has just fetched come back from the vector L1 in ~100 clocks -- sooner than a fragment read queued behind two
            // blocks' LDS traffic: the reads must have RETURNED before this wave lets the others through.  (Found by tests/test_determinism_gpu.py on the SDXL b4 plan: one
            // generation in a few differed in the last bits; the 2000-launch soak of single shapes did not hit it.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tt_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        half(1, NEXT_, 0, s ^ 1, ISSUE_, s);
    };

    auto mainloop = [&](int nk) __attribute__((always_inline)) {
        issue_tile(0); issue_tile(1);
        tt_wait_vmcnt<NS>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int g = 0; g < NS; ++g) read_one(0, g, 0);
        int t = 0;
        for (; t + 2 < nk; ++t) body(t & 1, std::true_type{}, std::true_type{});
        body(t & 1, std::true_type{}, std::false_type{}); ++t;
        body(t & 1, std::false_type{}, std::false_type{});
    };
    mainloop(nkt);

    if (p.tbuf && tid == 0) p.tbuf[(long)v * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue: acc[i][c][e] = C[wrow0 + 16 i + l15][wcol0 + 16 c + 4 lg + e]
    const int wrow0 = m0 + wr * WM, wcol0 = n0 + wc * WN;
    f32x4 cb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cb[c] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wcol0 + c * 16 + 4 * lg) : f32x4{0.f, 0.f, 0.f, 0.f};
    auto epi_f16 = [&](_Float16* C16, long ldc16) __attribute__((always_inline)) {
        _Float16* rowp = C16 + (long)(wrow0 + l15) * ldc16 + wcol0 + 8 * (lg >> 1) + 16 * (lg & 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int c = 0; c + 1 < NC; c += 2) {
                const f32x4 v0 = acc[i][c] + cb[c], v1 = acc[i][c + 1] + cb[c + 1];
                const f16x4 h0 = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3]};
                const f16x4 h1 = {(_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                const u32x2 a = __builtin_bit_cast(u32x2, h0), b = __builtin_bit_cast(u32x2, h1);
                const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                *reinterpret_cast<u32x4*>(rowp + (long)i * 16 * ldc16 + c * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
            const f32x4 vl = acc[i][NC - 1] + cb[NC - 1];      // odd block count: the last block keeps its 8-byte pieces
            const f16x4 hl = {(_Float16)vl[0], (_Float16)vl[1], (_Float16)vl[2], (_Float16)vl[3]};
            *reinterpret_cast<f16x4*>(C16 + (long)(wrow0 + l15 + i * 16) * ldc16 + wcol0 + (NC - 1) * 16 + 4 * lg) = hl;
        }
    };
    if constexpr (EPI == TT_F16) {
        epi_f16(p.C16, p.ldc16);
    } else if constexpr (EPI == TT_F32 || EPI == TT_F32_RES) {
        constexpr bool RES = EPI == TT_F32_RES;
        float* rowp = p.C32 + (long)(wrow0 + l15) * p.ldc32 + wcol0 + 4 * lg;
        f32x4 rr[RES ? NI : 1][RES ? NC : 1];
        if constexpr (RES) {      // the whole residual tile of the lane in one batch (80 registers the fragments no longer need)
            const float* resp = p.resid + (long)(wrow0 + l15) * p.ldr + wcol0 + 4 * lg;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c) rr[i][c] = *reinterpret_cast<const f32x4*>(resp + (long)i * 16 * p.ldr + c * 16);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 v4 = acc[i][c] + cb[c];
                if constexpr (RES) v4 += rr[i][c];
                *reinterpret_cast<f32x4*>(rowp + (long)i * 16 * p.ldc32 + c * 16) = v4;
            }
    } else {
        // *_LN: fp32 output (+ residual) AND the LayerNorm of the finished rows as fp16 (gemm_pp.hpp epi_ln with nbn = N / 160 partner tiles and 2 wave columns)
        constexpr bool RES = EPI == TT_F32_RES_LN;
        f32x4 rr[RES ? NI : 1][RES ? NC : 1];
        if constexpr (RES) {
            const float* resp = p.resid + (long)(wrow0 + l15) * p.ldr + wcol0 + 4 * lg;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c) rr[i][c] = *reinterpret_cast<const f32x4*>(resp + (long)i * 16 * p.ldr + c * 16);
        }
        float mean_w[NI], m2_w[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float s1 = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 v4 = acc[i][c] + cb[c];
                if constexpr (RES) v4 += rr[i][c];
                acc[i][c] = v4;
                s1 += (v4[0] + v4[1]) + (v4[2] + v4[3]);
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            const float mu = s1 * (1.0f / WN);
            float qq = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 d = acc[i][c] - mu;
                qq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            qq += __shfl_xor(qq, 16, 64); qq += __shfl_xor(qq, 32, 64);
            mean_w[i] = mu; m2_w[i] = qq;
        }
        f32x2* red = reinterpret_cast<f32x2*>(smem + RING);          // [wave row][64 rows][2 wave columns]
        if (lg == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) red[(wr * 64 + i * 16 + l15) * 2 + wc] = f32x2{mean_w[i], m2_w[i]};
        }
        // the ds_writes above must have LANDED before this wave lets the others through (a raw s_barrier waits for nothing): with a second block hammering the CU's LDS port a
        // reader behind the barrier could overtake them and combine a stale pair -- one (tile, wave row)'s partial wrong, hence the LayerNorm of 64 rows wrong in ALL partner tiles
        // (round 5, tools/tt_ln_diag.py; it showed once in 30 .. 300 launches on grids of more than one block per CU, never with one)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto chan = [&](float& n, float& mu, float& m2, float nb, float mub, float m2b) __attribute__((always_inline)) {
            const float d = mub - mu, nn = n + nb;
            mu += d * (nb / nn); m2 += m2b + d * d * (n * nb / nn); n = nn;
        };
        float mean_t[NI], m2_t[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x2* e = red + (wr * 64 + i * 16 + l15) * 2;
            float n = (float)WN, mu = e[0][0], m2 = e[0][1];
            chan(n, mu, m2, (float)WN, e[1][0], e[1][1]);
            mean_t[i] = mu; m2_t[i] = m2;
        }
        // ---- the partner tiles' row statistics (round 6: no ticket, no counters -- the records carry their own validity).  Every tile writes, per row, ONE 16-byte record
        // {mean, tag, m2, tag} = two self-tagged 8-byte granules (write-through store), tag = the tag of this tile's records of the op's previous launch + 1 (read at kernel
        // entry, above); the scratch is the op's own (mlblock.c), so a record with the right tag can only be this launch's.  One wave per wave row loads the partners' records with agent-scope loads UNTIL both tags of all of them match (bounded), hands them to
        // the other waves through LDS, and everything else is as before: same fp32 values, combined in tile order.  Nothing to reset, nothing to clear after a give-up.
        const int nbn = p.nbn;
        const unsigned tag = ln_epoch + 1u;
        u32x4* gws = reinterpret_cast<u32x4*>(p.ln_ws) + ((long)bmi * nbn * BM);           // [tile column][128 rows] records
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)gws, 0, nbn * BM * 16, 0x00020000);
        if (wc == 0 && lg == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const u32x4 rec = {__builtin_bit_cast(unsigned, mean_t[i]), tag, __builtin_bit_cast(unsigned, m2_t[i]), tag};
                __builtin_amdgcn_raw_buffer_store_b128(rec, rs, (bni * BM + wr * 64 + i * 16 + l15) * 16, 0, 16);      // sc1: write-through
            }
        }
        float* c32b = p.C32 + (long)(wrow0 + l15) * p.ldc32 + wcol0 + 4 * lg;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(c32b + (long)i * 16 * p.ldc32 + c * 16) = acc[i][c];      // the fp32 output goes out while the partners' records arrive
        // (in the ring's LDS: every wave passed its last fragment reads before the barrier above, and this loop leaves no LDS-DMA in flight)
        f32x2* red2 = reinterpret_cast<f32x2*>(smem);                                      // [128 rows][8 tiles]
        if (wc == 0) {
            // lane (l15, lg): partner tiles lg and lg + 4, rows 16 i + l15 of this wave row.  (This tile's own record comes from registers.)
            const bool act0 = lg < nbn, act1 = lg + 4 < nbn;
            const bool own0 = lg == bni, own1 = lg + 4 == bni;
            u32x4 rec0[NI], rec1[NI];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                asm volatile("" ::: "memory");      // the record loads below are plain (readonly) buffer loads to the compiler: without this it hoists them out of the poll loop
                if (act0 && !own0) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) rec0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lg * BM + wr * 64 + i * 16 + l15) * 16, 0, 16);      // sc1: past the vector L1, to the XCD's L2
                }
                if (act1 && !own1) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) rec1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lg + 4) * BM + wr * 64 + i * 16 + l15) * 16, 0, 16);
                }
                if (act0 && !own0) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) ok = ok && rec0[i][1] == tag && rec0[i][3] == tag;
                }
                if (act1 && !own1) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) ok = ok && rec1[i][1] == tag && rec1[i][3] == tag;
                }
                if (__all(ok) || ++spins >= (1u << 20)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (spins >= (1u << 20) && lane == 0) __hip_atomic_store(p.ln_cnt + 8191, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sticky give-up word (mlctx_handoff_check)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x2* e = red2 + (wr * 64 + i * 16 + l15) * 8;
                if (act0) e[lg] = own0 ? f32x2{mean_t[i], m2_t[i]} : f32x2{__uint_as_float(rec0[i][0]), __uint_as_float(rec0[i][2])};
                if (act1) e[lg + 4] = own1 ? f32x2{mean_t[i], m2_t[i]} : f32x2{__uint_as_float(rec1[i][0]), __uint_as_float(rec1[i][2])};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the ds_writes have LANDED before the barrier lets the readers through
        __builtin_amdgcn_s_barrier();
        float mean_r[NI], rstd_r[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f32x2* e = red2 + (wr * 64 + i * 16 + l15) * 8;
            f32x2 tt = e[0];
            float n = (float)BN, mu = tt[0], m2 = tt[1];
            for (int b = 1; b < nbn; ++b) { tt = e[b]; chan(n, mu, m2, (float)BN, tt[0], tt[1]); }
            mean_r[i] = mu; rstd_r[i] = 1.0f / sqrtf(m2 / n + p.ln_eps);
        }
        f32x4 gm[NC], bt[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            gm[c] = *reinterpret_cast<const f32x4*>(p.ln_g + wcol0 + c * 16 + 4 * lg);
            bt[c] = *reinterpret_cast<const f32x4*>(p.ln_b + wcol0 + c * 16 + 4 * lg);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            _Float16* rowp = p.ln_y + (long)(wrow0 + l15 + i * 16) * p.ldln + wcol0 + 8 * (lg >> 1) + 16 * (lg & 1);
            auto y4 = [&](int c) __attribute__((always_inline)) {
                const f32x4 y = (acc[i][c] - mean_r[i]) * rstd_r[i] * gm[c] + bt[c];
                return f16x4{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
            };
#pragma unroll
            for (int c = 0; c + 1 < NC; c += 2) {
                const u32x2 a = __builtin_bit_cast(u32x2, y4(c)), b = __builtin_bit_cast(u32x2, y4(c + 1));
                const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                *reinterpret_cast<u32x4*>(rowp + c * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
            *reinterpret_cast<f16x4*>(p.ln_y + (long)(wrow0 + l15 + i * 16) * p.ldln + wcol0 + (NC - 1) * 16 + 4 * lg) = y4(NC - 1);
        }
    }
    if (p.tbuf && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); p.tbuf[(long)v * 4 + 2] = __builtin_amdgcn_s_memrealtime(); }
}

int g_tt_prio = 1;
unsigned long long* g_tt_tbuf = nullptr;      // 0: no priority classes (A/B)

}  // namespace

// problems the tile takes; returns the epilogue kind (TT_*) or 0.  `ncu`: CUs of the stream (partner tiles of a *_LN launch must be resident together)
extern "C" int mlsd_gemm_tt_eligible(const mlsd_gemm_args* a, int ncu)
{
    const bool linear = !a->conv || (a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad == 0 && !a->upsample && a->K == a->Cin);      // (a 1 x 1 convolution of an NHWC image IS a linear layer over its pixels)
    if (!linear || (a->K & 63) || a->K < 128 || (a->M % 128) || (a->N % 160)) return 0;
    if (a->rowbias || a->bias_m || a->colstats || a->act != MLSD_ACT_NONE || a->ksplit > 1) return 0;
    if (a->bias && ((uintptr_t)a->bias & 15)) return 0;
    if (a->C16 && !a->C32 && !a->resid && !a->ln_y16) return (!(a->ldc16 & 7) && !((uintptr_t)a->C16 & 15)) ? TT_F16 : 0;
    if (a->C32 && !a->C16) {
        if ((a->ldc32 & 3) || ((uintptr_t)a->C32 & 15)) return 0;
        if (a->resid && ((a->ldr & 3) || ((uintptr_t)a->resid & 15))) return 0;
        if (a->ln_y16) {
            if (!a->ln_gamma || !a->ln_beta || !a->ln_ws || !a->ln_cnt || (a->ldln & 7) || ((uintptr_t)a->ln_y16 & 15) || ((uintptr_t)a->ln_gamma & 15) || ((uintptr_t)a->ln_beta & 15)) return 0;
            // The N / 160 tiles of a row block wait for each other inside the launch: all of them must be resident together, which a grid of at most 2 blocks per CU guarantees
            // without any assumption on the dispatch order (larger grids work while blocks are dispatched in id order -- the soak runs them with MLSD_TT_LN_ANYGRID=1 -- but are
            // not taken).  Partner tiles are blocks 8 apart: one XCD (checked: XCC_ID == id % 8 for every block of 256 .. 1024-block grids, tools/gemm_tt_xcd_check.py), so the
            // write-through records and agent-scope loads meet in that XCD's L2.  At most 8 partner tiles (two per lane group of the gathering wave).
            static int anygrid = -1;      // MLSD_TT_LN_ANYGRID=1: also grids of more than 2 blocks per CU (tools/soak_r5.py)
            if (anygrid < 0) { const char* e = getenv("MLSD_TT_LN_ANYGRID"); anygrid = (e && *e == '1') ? 1 : 0; }
            if (a->N / 160 > 8 || a->ln_slot < 0 || a->ln_slot >= 8191 || ncu < 256 || (!anygrid && (long)(a->M / 128) * (a->N / 160) > 2L * ncu)) return 0;
            return a->resid ? TT_F32_RES_LN : TT_F32_LN;
        }
        return a->resid ? TT_F32_RES : TT_F32;
    }
    return 0;
}

extern "C" int mlsd_gemm_tt(const mlsd_gemm_args* a, void* stream, int ncu)
{
    const int epi = mlsd_gemm_tt_eligible(a, ncu);
    if (!epi) return mlsd_set_error(-1, "mlsd_gemm_tt: problem not eligible for the 128x160 two-tiles-per-CU kernel");
    TTP p;
    p.A = (const _Float16*)a->A; p.B = (const _Float16*)a->W_; p.lda = a->lda; p.ldb = a->ldb; p.M = a->M; p.N = a->N; p.K = a->K;
    p.bias = a->bias; p.resid = a->resid; p.ldr = a->ldr; p.C32 = a->C32; p.ldc32 = a->ldc32; p.C16 = (_Float16*)a->C16; p.ldc16 = a->ldc16;
    p.nbm = a->M / 128; p.nbn = a->N / 160; p.prio = g_tt_prio;
    p.ln_g = a->ln_gamma; p.ln_b = a->ln_beta; p.ln_eps = a->ln_eps; p.ln_y = (_Float16*)a->ln_y16; p.ldln = a->ldln; p.ln_ws = a->ln_ws; p.ln_cnt = a->ln_cnt; p.ln_slot = a->ln_slot; p.tbuf = g_tt_tbuf;
    const bool ln = epi >= TT_F32_LN;
    const size_t LDS = 2 * (size_t)(128 + 160) * 128 + (ln ? 2048 : 0);
    const dim3 grid(p.nbm * p.nbn), block(256);
    auto go = [&](auto kfn) -> int {
        static thread_local const void* attr_done[8]; static thread_local int n_done = 0;      // (one host call per kernel, not per launch)
        bool seen = false;
        for (int i = 0; i < n_done; ++i) seen |= attr_done[i] == (const void*)kfn;
        if (!seen) {
            MLSD_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 160) * 128 + 2048));
            if (n_done < 8) attr_done[n_done++] = (const void*)kfn;
        }
        hipLaunchKernelGGL(kfn, grid, block, LDS, (hipStream_t)stream, p);
        return mlsd_check_launch("gemm_tt_kernel");
    };
    switch (epi) {
    case TT_F16: return go(gemm_tt_kernel<TT_F16>);
    case TT_F32: return go(gemm_tt_kernel<TT_F32>);
    case TT_F32_RES: return go(gemm_tt_kernel<TT_F32_RES>);
    case TT_F32_LN: return go(gemm_tt_kernel<TT_F32_LN>);
    default: return go(gemm_tt_kernel<TT_F32_RES_LN>);
    }
}

extern "C" MLSD_API void mlsd_gemm_tt_set_prio(int on) { g_tt_prio = on ? 1 : 0; }
extern "C" MLSD_API void mlsd_gemm_tt_set_trace(void* buf) { g_tt_tbuf = (unsigned long long*)buf; }      /* 4 x uint64 per block, NULL = off */
