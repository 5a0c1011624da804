// 256x256x64 persistent "ping-pong" GEMM / implicit-GEMM conv for gfx950 — included by gemm_conv.hip (inside
// its anonymous namespace; uses GemmP, wait_vmcnt, g_zero_page and the activation helpers).
//
// 8 waves = 2 groups (wave rows wr = 0,1) x 4 wave columns; a wave owns a 128x64 output = 4 quadrants of
// 64x32, accumulated with v_mfma_f32_16x16x32_f16 (128 accumulator registers).  One wave of each group
// sits on every SIMD, and the groups run ONE BARRIER APART: while group 0 issues the 16 MFMAs of a phase
// (s_setprio 1), group 1 issues the LDS fragment reads and the LDS-DMA of its next phase, and vice versa.
// (Structure after cdna_hip_programming.md "The 256^2 8-phase template"; scheduling re-derived for this
// kernel's staging units and swizzle.)
//
//   K tile = 4 phases, phase P computes quadrant (qa,qb) over the tile's K = 64:
//       P1 (a0,b0)   P2 (a0,b1)   P3 (a1,b1)   P4 (a1,b0)        fragments: a* 8 x ds_read_b128, b* 4
//   LDS: 2 stages x (A 256 rows | B 256 rows) x 128 B (same XOR swizzle as the other tiles), filled in
//   four UNITS of 128 rows (16 KiB = 2 LDS-DMA per thread), cut the way the phases consume them:
//       U1 = A rows of every wave's a0,  U2 = B rows of every b0,  U3 = B rows of b1,  U4 = A rows of a1
//   Issue schedule (one unit per phase, s = position in the K-tile stream):
//       P1: U2(s+1)   P2: U3(s+1)   P3: U4(s+1)   P4: U1(s+2)
//   => every unit has >= 3 phases of flight before the counted wait that retires it, and is restaged >= 3
//      phases after its last fragment read (WAR).
//   RAW rule (a reader group is one barrier behind/ahead of the other): the unit read in phase g must be
//   retired by EVERY wave's counted vmcnt in phase g-1, before that phase's first barrier:
//       wait vmcnt(6) in P4 retires U1,U2(s+1); in P1 retires U3(s); in P2 retires U4(s).   (6 = the three
//       younger units x 2 instructions; never 0 inside the loop.  vmcnt also counts the epilogue's stores,
//       which are older than any unit waited for, so the count stays valid across a tile seam.)
//
// PERSISTENT: the grid is min(tiles, 256) blocks; block b walks tiles b, b+G, b+2G, ... and the K tiles of
// consecutive output tiles form ONE stream: at the end of an output tile the first units of the next one are
// already in flight (both ring stages are busy), the epilogue stores straight from the accumulators (transposed
// product: a lane holds 4 consecutive columns) without touching LDS or draining anything, and the next main
// loop starts on landed data.  With K = 1280 (20 K tiles per output tile) the exposed prologue + epilogue was a third of
// the tile time.  Units past the end of the stream are staged from the zero page (uniform counts).
struct PPTile { int m0, n0; };

__device__ __forceinline__ PPTile pp_tile_coords(const GemmP& p, int v)
{
    // XCD-aware bijective remap of the virtual block index + column panels (as gemm_kernel)
    const int nblk = p.nbm * p.nbn;
    const int q = nblk >> 3, r = nblk & 7, x = v & 7, j = v >> 3;
    int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
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
    return PPTile{bm * 256, bn * 256};
}

// Linear only (plain row-major A), M a multiple of 128, N of 64, K a multiple of 64 and >= 192 (3 K tiles: U1, two tiles ahead, must enter output
// tile ti+1 while the MFMAs are in tile ti): the per-phase staging code is two LDS-DMA
// instructions on running row pointers and nothing else (out-of-range rows are CLAMPED to the last valid row
// instead of zero-filled: their products land in output rows/columns that are never stored).
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmP p)
{
    constexpr int BK = 64;
    constexpr int STAGE = 512 * BK * 2;            // 64 KiB: A 256 rows | B 256 rows
    constexpr int BOFF = 256 * BK * 2;             // B tile offset inside a stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations (M0) stay on the SALU
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nblk = p.nbm * p.nbn;
    const int G = gridDim.x;
    const int ntile = (nblk - (int)blockIdx.x + G - 1) / G;      // output tiles of this block (>= 1)
    const int nkt = p.K / BK;
    const int S = ntile * nkt;                                   // K tiles in this block's stream

    // ---- staging assignment.  A unit = 128 tile rows = 2 x (8 waves x 8 rows); this thread stages, for every
    // 64-row block it touches, row srow of the block and the LOGICAL chunk whose swizzled slot is lane&7.
    const int srow = wave * 8 + (lane >> 3);                 // 0..63
    const int sc = (lane & 7) ^ ((srow >> 1) & 7);           // every row this thread stages is = srow mod 16

    // One iterator per unit sequence: U1 runs two K tiles ahead of the MFMAs, U2..U4 one.  Each holds two running
    // row pointers (this thread's chunk of the current K tile) for the output tile it is staging for; all of them
    // enter output tile ti+1 while the MFMAs are still in tile ti, so "the next tile" is one shared coordinate pair.
    struct Seq { int kt, par; const _Float16* ptr[2]; };
    Seq sa[2];    // [0] = U1 (tile rows srow + {0,128}), [1] = U4 (rows srow + {64,192})
    Seq sb[2];    // [0] = U2, [1] = U3
    PPTile tnext = pp_tile_coords(p, (int)blockIdx.x);       // coordinates the sequences use at their next tile entry
    PPTile tcur = tnext;                                     // tile of the MFMAs / next epilogue

    auto enter_A = [&](Seq& s, int second) __attribute__((always_inline)) {
        s.kt = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = min(tnext.m0 + srow + 64 * (it * 2 + second), p.M - 1);
            s.ptr[it] = p.A + (long)m * p.lda + sc * 8;
        }
    };
    auto enter_B = [&](Seq& s, int second) __attribute__((always_inline)) {
        s.kt = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int u = it * 64 + srow;
            const int n = min(tnext.n0 + (u >> 5) * 64 + (u & 31) + second * 32, p.N - 1);   // unit row u -> tile row
            s.ptr[it] = p.B + (long)n * p.ldb + sc * 8;
        }
    };
    auto issue_A = [&](Seq& s, int second) __attribute__((always_inline)) {
        unsigned char* stage = smem + s.par * STAGE;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned char* dst = stage + ((it * 2 + second) * 64 + wave * 8) * (BK * 2);   // wave-uniform: 8 rows, lane-linear
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s.ptr[it],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            s.ptr[it] += BK;
        }
        s.par ^= 1;
        if (++s.kt == nkt) enter_A(s, second);
    };
    auto issue_B = [&](Seq& s, int second) __attribute__((always_inline)) {
        unsigned char* stage = smem + s.par * STAGE + BOFF;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r0 = (it * 2 + (wave >> 2)) * 64 + (wave & 3) * 8 + second * 32;    // first of the wave-instruction's 8 rows
            unsigned char* dst = stage + r0 * (BK * 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s.ptr[it],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            s.ptr[it] += BK;
        }
        s.par ^= 1;
        if (++s.kt == nkt) enter_B(s, second);
    };

    // ---- fragment addressing (16x16x32: lane = (row l15, k group lg)); rows are = l15 mod 16, so the swizzle
    // term is lane-constant: slot(ks) = ((4*ks + lg) ^ (l15 >> 1)) = c0 ^ (4*ks)
    const int c0 = lg ^ (l15 >> 1);
    const int fa = (wr * 128 + l15) * (BK * 2);                // + (qa*64 + i*16) rows
    const int fb = BOFF + (wc * 64 + l15) * (BK * 2);          // + (qb*32 + j*16) rows
    const int fk0 = c0 << 4, fk1 = (c0 ^ 4) << 4;

    f32x4 acc[2][2][4][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    f16x8 af[4][2], bf[2][2][2];

    auto read_A = [&](const unsigned char* stage, int qa) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* r = stage + fa + (qa * 64 + i * 16) * (BK * 2);
            af[i][0] = *reinterpret_cast<const f16x8*>(r + fk0);
            af[i][1] = *reinterpret_cast<const f16x8*>(r + fk1);
        }
    };
    auto read_B = [&](const unsigned char* stage, int qb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* r = stage + fb + (qb * 32 + j * 16) * (BK * 2);
            bf[qb][j][0] = *reinterpret_cast<const f16x8*>(r + fk0);
            bf[qb][j][1] = *reinterpret_cast<const f16x8*>(r + fk1);
        }
    };

    // ---- epilogue of one output tile, straight from the accumulators.  The MFMAs compute the TRANSPOSED product
    // (B fragment as the first operand), so acc[qa][qb][i][j][e] = C[row qa*64 + i*16 + l15][col qb*32 + j*16 + 4*lg + e]
    // inside the wave's 128x64: a lane holds 4 CONSECUTIVE COLUMNS of one row -> 16-byte fp32 / 8-byte fp16 accesses with
    // no LDS transposition (the LDS pass of the other tiles costs ~9 instructions per element; here ~1), and GEGLU's
    // value (qb = 0) and gate (qb = 1) of one output element sit in the same lane.
    // The epilogue is straight-line code instantiated for 32 accumulator tiles, executed once per output tile: it must
    // stay SMALL (a first version with a per-element activation switch and scalar fallbacks was 100 KB of code and
    // cost 12 us per tile in instruction fetch).  Hence: 16-byte-aligned layouts only (the launcher sends anything else
    // to the LDS-transposing tiles) and ONE activation formula  y = x * sigmoid(x * (c1 + c3 x^2))  for SiLU (1, 0),
    // quick-GELU (1.702, 0) and tanh-GELU (2c, 2c*0.044715), or  y = max(x, lo)  for none (lo = -inf) / ReLU (lo = 0).
    const bool geglu = p.act == MLSD_ACT_GEGLU;
    const bool act_sig = p.act == MLSD_ACT_SILU || p.act == MLSD_ACT_GELU || p.act == MLSD_ACT_GELU_QUICK;
    const float act_c1 = p.act == MLSD_ACT_SILU ? 1.0f : (p.act == MLSD_ACT_GELU_QUICK ? 1.702f : 2.0f * 0.7978845608028654f);
    const float act_c3 = p.act == MLSD_ACT_GELU ? 2.0f * 0.7978845608028654f * 0.044715f : 0.0f;
    const float act_lo = p.act == MLSD_ACT_RELU ? 0.0f : -3.0e38f;
    auto act4 = [&](f32x4 v) __attribute__((always_inline)) {
        if (act_sig) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * fast_sigmoid(v[e] * fmaf(act_c3 * v[e], v[e], act_c1));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], act_lo);
        }
        return v;
    };
    // Launcher guarantees (else the LDS-transposing tiles run): M a multiple of 128 and N of 64 (a wave's 128x64 block is
    // entirely inside or outside: one wave-uniform test, no per-lane guards), 16-byte aligned rows, and rows_per_batch a
    // multiple of 256 (one row-bias vector per output tile).
    // one 16-row block (qa, i) of the wave's tile; called with compile-time indices (a pragma-unrolled loop over the
    // whole epilogue exceeds the unroller's size limit and would push the accumulators to scratch)
    f32x4 cb[2][2];       // bias (+ row bias) of the lane's 4 column groups, loaded once per output tile
    auto epi_rows = [&](auto QA, auto I, int wrow0, int wcol0) __attribute__((always_inline)) {
        constexpr int qa = decltype(QA)::value, i = decltype(I)::value;
        const int m = wrow0 + qa * 64 + i * 16 + l15;
        const float bm = p.biasm ? p.biasm[m] : 0.f;
        if (!geglu) {
            const long col = wcol0 + 4 * lg;
            const float* rrow = p.resid ? p.resid + (long)m * p.ldr + col : nullptr;
            float* c32 = p.C32 ? p.C32 + (long)m * p.ldc32 + col : nullptr;
            _Float16* c16 = p.C16 ? p.C16 + (long)m * p.ldc16 + col : nullptr;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int o = qb * 32 + j * 16;          // column offset inside the wave's 64 (an immediate)
                    f32x4 v = acc[qa][qb][i][j] + cb[qb][j];
                    v += bm;
                    f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                    if (rrow) rs = *reinterpret_cast<const f32x4*>(rrow + o);
                    if (p.act_post) v += rs;
                    v = act4(v);
                    if (!p.act_post) v += rs;
                    if (c32) *reinterpret_cast<f32x4*>(c32 + o) = v;
                    if (c16) {
                        f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        *reinterpret_cast<f16x4*>(c16 + o) = h;
                    }
                }
            }
        } else {
            const long col = (wcol0 >> 6) * 32 + 4 * lg;     // value | gate column blocks of 32 are interleaved
            const float* rrow = p.resid ? p.resid + (long)m * p.ldr + col : nullptr;
            float* c32 = p.C32 ? p.C32 + (long)m * p.ldc32 + col : nullptr;
            _Float16* c16 = p.C16 ? p.C16 + (long)m * p.ldc16 + col : nullptr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = j * 16;
                const f32x4 a4 = acc[qa][0][i][j] + cb[0][j], g4 = acc[qa][1][i][j] + cb[1][j];
                f32x4 v;
                v[0] = a4[0] * gelu_tanh_f(g4[0]); v[1] = a4[1] * gelu_tanh_f(g4[1]);
                v[2] = a4[2] * gelu_tanh_f(g4[2]); v[3] = a4[3] * gelu_tanh_f(g4[3]);
                if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + o);
                if (c32) *reinterpret_cast<f32x4*>(c32 + o) = v;
                if (c16) {
                    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4*>(c16 + o) = h;
                }
            }
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int wrow0 = tcur.m0 + wr * 128, wcol0 = tcur.n0 + wc * 64;
        if (wrow0 >= p.M || wcol0 >= p.N) return;             // M % 128 == 0, N % 64 == 0: a wave's 128x64 is all in or all out
        const float* rbias = p.rowbias ? p.rowbias + (long)(tcur.m0 / p.rows_per_batch) * p.ldrb : nullptr;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = wcol0 + qb * 32 + j * 16 + 4 * lg;
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) c = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (rbias) c += *reinterpret_cast<const f32x4*>(rbias + n);
                cb[qb][j] = c;
            }
        using std::integral_constant;
        epi_rows(integral_constant<int, 0>{}, integral_constant<int, 0>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 0>{}, integral_constant<int, 1>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 0>{}, integral_constant<int, 2>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 0>{}, integral_constant<int, 3>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 1>{}, integral_constant<int, 0>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 1>{}, integral_constant<int, 1>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 1>{}, integral_constant<int, 2>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 1>{}, integral_constant<int, 3>{}, wrow0, wcol0);
    };

    // ---- prologue: U1..U4 of stream position 0 and U1 of position 1 in flight; U1(0), U2(0) retired and visible
    sa[0].par = sa[1].par = sb[0].par = sb[1].par = 0;
    enter_A(sa[0], 0); enter_A(sa[1], 1); enter_B(sb[0], 0); enter_B(sb[1], 1);
    tcur = tnext;
    tnext = pp_tile_coords(p, (int)blockIdx.x + (ntile > 1 ? G : 0));   // past the end: any valid tile (staged, never read)
    issue_A(sa[0], 0); issue_B(sb[0], 0); issue_B(sb[1], 1); issue_A(sa[1], 1); issue_A(sa[0], 0);
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one barrier behind group 0 from here on

#define MLSD_PP_PHASE(QA, QB, LOAD_A, LOAD_B, ISSUE, WAIT)                                                   \
    {                                                                                                        \
        if (LOAD_B) read_B(stage, QB);                                                                       \
        if (LOAD_A) read_A(stage, QA);                                                                       \
        ISSUE;                                                                                               \
        if (WAIT) wait_vmcnt<6>();                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
                    acc[QA][QB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[QB][j][ks], af[i][ks], acc[QA][QB][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

    int kt = 0, ti = 0;
    for (int s = 0; s < S; ++s) {
        const unsigned char* stage = smem + (s & 1) * STAGE;
        MLSD_PP_PHASE(0, 0, true, true, issue_B(sb[0], 0), true)
        MLSD_PP_PHASE(0, 1, false, true, issue_B(sb[1], 1), true)
        MLSD_PP_PHASE(1, 1, true, false, issue_A(sa[1], 1), false)
        MLSD_PP_PHASE(1, 0, false, false, issue_A(sa[0], 0), true)
        if (++kt == nkt) {
            // seam: the next output tile's first units are in flight / landed; nothing is drained.  The groups
            // rejoin (group 0's extra barrier pairs with group 1's last one) so that all 8 waves run their
            // epilogues side by side, then group 1 drops one barrier behind again (equal barrier counts).
            if (wr == 0) __builtin_amdgcn_s_barrier();
            if (!(p.dbg & 1)) epilogue();
            zero_acc();
            kt = 0; ++ti;
            tcur = tnext;                                  // every sequence has entered tile ti by now (K >= 3 K tiles)
            tnext = pp_tile_coords(p, (int)blockIdx.x + (ti + 1 < ntile ? ti + 1 : 0) * G);
            if (s + 1 < S && wr == 1) __builtin_amdgcn_s_barrier();
        }
    }
#undef MLSD_PP_PHASE
    wait_vmcnt<0>();                               // the zero-page tail stages still target this block's LDS
}
