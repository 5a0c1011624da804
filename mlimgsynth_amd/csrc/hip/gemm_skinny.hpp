// Skinny-M weight-streaming GEMM / implicit-GEMM conv for gfx950 (round 4) -- included by gemm_conv.hip inside its anonymous
// namespace (uses GemmP, lds_off, wait_vmcnt, g_zero_page).
//
//   C[M,N] = A[M,K] . W[N,K]^T   with M <= 128: the 8x8 level of SD1.5 at batch 1 (cond + uncond = 128 rows), the time-embedding
//   projections (M = 2).  Replaces ggml_mul_mat / ggml_conv_2d (/root/reference/src/mlblock_nn.c:16-55) at the sizes of
//   /root/reference/src/unet.c:21-39 where the weights (29.5 MB for a 1280 -> 1280 3x3 conv) are the only bytes that matter:
//   3.8 GFLOP against 29.5 MB = 128 FLOP/B, far under the ridge (310): the launch is an HBM stream of the weights.
//
// What the general tiles do wrong here (64x128 tile, split-K over 13 slices: 21 us = 1.4 TB/s, PMC 2.9x the algorithmic bytes):
// two row tiles fetch every weight byte twice, the weights go through the LDS ring behind one barrier per K tile (one K tile in
// flight per block), and LDS carries 3 bytes per weight byte.
//
// Here:
//   * a block holds ALL M rows: every weight byte is read by exactly one block, once;
//   * the weights never touch LDS: lane (n = l & 15, g = l >> 4) loads 2 x 16 contiguous bytes of row n straight into the registers
//     that feed the MFMA (four lanes cover a full 128-byte line of the row per 64-wide K step), PF = 3 K steps ahead -- 6 independent
//     16-byte loads in flight per lane, two blocks per CU (48 KB of weights in flight per CU), no barrier between a load and its use;
//   * only the (small, L2-resident) activations are staged through LDS, by LDS-DMA, in a ring of PF + 1 stages of 128 rows x 64: vmcnt retires loads in
//     order, so the activation stage of K step s is issued PF steps ahead too, immediately BEFORE the weight loads of step s -- waiting for "step kt+1 landed"
//     then leaves the PF - 1 younger steps (activations and weights) in flight instead of draining the weight stream every iteration;
//   * K is permuted consistently in both operands so that a lane's bytes are contiguous: MFMA j of a K step contracts
//     k = 16 g + 8 j + (0..7) for lane group g in both fragments (any permutation of k is legal as long as A and W agree);
//   * grid = (N / 64) x S: block (bn, z) owns weight columns [64 bn, 64 bn + 64) x K slice z -- a disjoint slab of W --
//     and writes its raw fp32 partial tile to slice z of the workspace; the slices are added in FIXED order by splitk_reduce
//     (second dispatch: measured cheaper than any in-launch reduction, profiles/NOTES.md), which also applies the epilogue.
//
// v_mfma_f32_16x16x32_f16 computes the transposed product (weights as the first operand), so acc[rg][e] = C[16 rg + (l & 15)][n0 + 16 wave + 4 g + e]:
// a lane stores 4 consecutive columns (16 bytes).
//
// Algorithmic bytes per launch: N K 2 (weights) + A once + S M N 4 (partials, written and read once).
template <int I, int N, class F>
__device__ __forceinline__ void skinny_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); skinny_static_for<I + 1, N>(f); }
}

constexpr int SKINNY_MAXG = 8;          // groups of PF + 1 K steps a slice may have (gemm_skinny_kernel is straight-line code)
template <int G, int NST, class F>
__device__ __forceinline__ void skinny_groups(int nkt, F&& f)
{
    if constexpr (G < SKINNY_MAXG) {
        if (G * NST < nkt) { f(); skinny_groups<G + 1, NST>(nkt, f); }
    }
}

template <bool CONV, int RG, int PF, int DBG = 0>      // DBG (timing-only builds: wrong results): 1 = no activation staging / LDS reads / MFMAs (the weight stream alone)
// RG: 16-row groups of the tile (8: M <= 128, 4: M <= 64, 1: M <= 16); PF: K steps in flight per thread beyond the one consumed
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmP p)
{
    constexpr int BM = RG * 16, BK = 64, RB = BK * 2;
    constexpr int NST = PF + 1;                             // activation ring
    constexpr int ROWS = BM < 32 ? 32 : BM;                 // (a wave-instruction fills 8 rows: 4 waves = 32 rows minimum)
    constexpr int A_IT = ROWS * 8 / 256;                    // LDS-DMA instructions per thread per stage (16-byte chunks: ROWS x 8)
    constexpr int GRP = A_IT + 2;                           // vector-memory instructions a thread issues per K step (activation DMA + 2 weight loads)
    constexpr int STAGE = ROWS * RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [NST][ROWS][64] halfs, XOR-swizzled like the other tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 64 + wave * 16;             // this wave's 16 weight rows (= output columns)
    const int nkt_all = p.K / BK;
    const int kt0 = blockIdx.y * p.kt_per;
    const int nkt = min(p.kt_per, nkt_all - kt0);
    if (nkt <= 0) return;                                   // (never: the launcher sizes the grid)

    // ---- weight stream: lane (l15, lg) reads row n0 + l15, bytes [128 kt + 32 lg, +32) of K step kt
    const int nrow = min(n0 + l15, p.N - 1);                // (rows past N: clamped, their columns are never stored)
    const _Float16* wp = p.B + (long)nrow * p.ldb + (long)kt0 * BK + lg * 16;
    f16x8 wq[NST][2];

    // ---- activation staging (LDS-DMA).  Thread t fills slot t & 7 of row t >> 3 (+ 32 i): it fetches the LOGICAL chunk whose swizzled
    // position that is.  Conv: a K step lies inside one filter tap (Cin % 64 == 0): tap and channel offset are block-uniform.
    const int sr = tid >> 3;
    const _Float16* zsrc = reinterpret_cast<const _Float16*>(g_zero_page);
    const _Float16* arow[A_IT];
    int amask[A_IT];
    int a_kh = 0, a_kw = 0, a_cin = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = sr + i * 32;
        const int ch = ((tid & 7) ^ ((r >> 1) & 7)) * 8;
        arow[i] = zsrc; amask[i] = 0;
        if (r < BM && r < p.M) {
            if constexpr (CONV) {
                const int ohw = p.OH * p.OW;
                const int img = r / ohw, rem = r - img * ohw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
                int mk = 0;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw)
                        if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mk |= 1 << (kh * p.KW + kw);
                amask[i] = mk;
                arow[i] = p.A + ((long)img * p.H * p.W + (long)ih0 * p.W + iw0) * p.lda + ch;
            } else {
                amask[i] = 1;
                arow[i] = p.A + (long)r * p.lda + ch + (long)kt0 * BK;
            }
        }
    }
    if constexpr (CONV) {
        const int k0 = kt0 * BK, tap = k0 / p.Cin;
        a_cin = k0 - tap * p.Cin; a_kh = tap / p.KW; a_kw = tap - a_kh * p.KW;
    }
    // one K step's loads: the activation stage into ring slot `slot` (EXACTLY A_IT instructions), then the two weight loads into register slot `slot`.
    // UNIFORM by construction: steps past the end of the slice (s >= nkt) issue the same instructions on the 64-byte zero page, and the loop below runs whole groups
    // of NST steps -- the extra steps multiply zeros.  (With a conditional issue the compiler's own wait insertion has to assume the shortest path and puts
    // s_waitcnt vmcnt(0) in front of the first MFMA of every step: the stream then drains every iteration.)
    int s_issue = 0;                                        // K step the next issue_step call stages
    auto issue_step = [&](auto SLOT) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
        const bool real = s_issue < nkt;
        __builtin_amdgcn_sched_barrier(0);
        unsigned char* As = smem + slot * STAGE;
        const long toff = CONV ? ((long)a_kh * p.W + a_kw) * p.lda + a_cin : 0;
        const int tbit = (CONV ? 1 << (a_kh * p.KW + a_kw) : 1) & (real ? -1 : 0);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const _Float16* src = (amask[i] & tbit) ? arow[i] + toff : zsrc;
            unsigned char* dst = As + (wave * 8 + i * 32) * RB;                  // wave-uniform: 8 rows, lane-linear
            if constexpr (DBG & 1) src = zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            if constexpr (!CONV) { if (amask[i]) arow[i] += BK; }
        }
        if constexpr (CONV) { a_cin += BK; if (a_cin == p.Cin) { a_cin = 0; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } } }
        const _Float16* ws = real ? wp : zsrc;
        __builtin_amdgcn_sched_barrier(0);                  // the ISSUE ORDER is the contract of the counted waits: [A stage s][W step s] per step, steps in order
        wq[slot][0] = *reinterpret_cast<const f16x8*>(ws);  // (left free, the scheduler hoists all DMA of the prologue above all weight loads)
        wq[slot][1] = *reinterpret_cast<const f16x8*>(ws + 8);
        __builtin_amdgcn_sched_barrier(0);
        wp += BK;
        ++s_issue;
    };
    static_assert((PF - 1) * GRP < 64, "vmcnt immediate");

    f32x4 acc[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    using std::integral_constant;
    // prologue: K steps 0 .. PF-1 in flight (ring / register slots 0 .. PF-1)
    skinny_static_for<0, PF>([&](auto S) __attribute__((always_inline)) { issue_step(S); });
    wait_vmcnt<(PF - 1) * GRP>();                           // step 0 landed

    // main loop in whole groups of NST steps (ring and register slots are compile-time: step kt lives in slot kt % NST)
    auto step = [&](auto SLOT) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
        __builtin_amdgcn_s_barrier();                       // every thread's DMA of this step has landed (its own counted wait) -> visible to all; the previous step's readers are done
        const f16x8 w0 = wq[slot][0], w1 = wq[slot][1];
        const unsigned char* As = smem + slot * STAGE;
#pragma unroll
        for (int rg = 0; rg < ((DBG & 1) ? 1 : RG); ++rg) {
            const int row = rg * 16 + l15;
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(As + lds_off<64>(row, lg * 2));
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(As + lds_off<64>(row, lg * 2 + 1));
            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, a0, acc[rg], 0, 0, 0);
            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, a1, acc[rg], 0, 0, 0);
        }
        // PF steps ahead, into the slot of the PREVIOUS step (its readers passed the barrier above).  Issued AFTER this step's MFMAs were placed: with the issue in
        // front of them the compiler's own wait for this step's weight registers has 7 x 6 younger loads to count and comes out as vmcnt(0) (seen in the ISA).
        issue_step(integral_constant<int, (slot + PF) % NST>{});
        wait_vmcnt<(PF - 1) * GRP>();                       // the NEXT step has landed; the PF - 1 steps behind it stay in flight
    };
    // No loop: up to SKINNY_MAXG groups of NST steps as straight-line code behind forward branches (the launcher keeps a slice within SKINNY_MAXG * NST steps).  With a
    // back edge the compiler's wait insertion merges the prologue's and the previous iteration's pending loads at the loop header and drains the stream there
    // (s_waitcnt vmcnt(0) once per NST steps, measured in the ISA); straight-line code gets exact counts.
    skinny_groups<0, NST>(nkt, [&]() __attribute__((always_inline)) {
        skinny_static_for<0, NST>([&](auto S) __attribute__((always_inline)) { step(S); });
    });
    wait_vmcnt<0>();                                        // (the trailing zero-page stages still target this block's LDS)

    // ---- raw partial tile -> slice blockIdx.y of the workspace (row-major [M][N] like the other split-K tiles; splitk_reduce adds the slices and applies the epilogue)
    float* const C = p.C32 + (long)blockIdx.y * p.ws_stride;
    const int n = n0 + 4 * lg;
    if (n < p.N) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            const int m = rg * 16 + l15;
            if (m < p.M) *reinterpret_cast<f32x4*>(C + (long)m * p.ldc32 + n) = acc[rg];
        }
    }
}
