// 256x256x64 "ping-pong" GEMM / implicit-GEMM conv for gfx950 — included by gemm_conv.hip (inside its
// anonymous namespace; uses GemmP, lds_off, wait_vmcnt, g_zero_page and the activation helpers).
//
// 8 waves = 2 groups (wave rows wr = 0,1) x 4 wave columns; a wave owns a 128x64 output = 4 quadrants of
// 64x32, accumulated with v_mfma_f32_16x16x32_f16 (128 accumulator registers).  One wave of each group
// sits on every SIMD, and the groups run ONE BARRIER APART: while group 0 issues the 16 MFMAs of a phase
// (s_setprio 1), group 1 issues the LDS fragment reads and the LDS-DMA of its next phase, and vice versa.
// (Structure after cdna_hip_programming.md "The 256^2 8-phase template"; scheduling re-derived for this
// kernel's staging units and swizzle.)
//
//   K tile kt = 4 phases, phase P computes quadrant (qa,qb) over the tile's K = 64:
//       P1 (a0,b0)   P2 (a0,b1)   P3 (a1,b1)   P4 (a1,b0)        fragments: a* 8 x ds_read_b128, b* 4
//   LDS: 2 stages x (A 256 rows | B 256 rows) x 128 B (same XOR swizzle as the other tiles), filled in
//   four UNITS of 128 rows (16 KiB = 2 LDS-DMA per thread), cut the way the phases consume them:
//       U1 = A rows of every wave's a0,  U2 = B rows of every b0,  U3 = B rows of b1,  U4 = A rows of a1
//   Issue schedule (one unit per phase):  P1: U2(kt+1)  P2: U3(kt+1)  P3: U4(kt+1)  P4: U1(kt+2)
//   => every unit has >= 3 phases of flight before the counted wait that retires it, and is restaged >= 3
//      phases after its last fragment read (WAR).
//   RAW rule (a reader group is one barrier behind/ahead of the other): the unit read in phase g must be
//   retired by EVERY wave's counted vmcnt in phase g-1, before that phase's first barrier:
//       wait vmcnt(6) in P4 retires U1,U2(kt+1); in P1 retires U3(kt); in P2 retires U4(kt).   (6 = the
//       three younger units x 2 instructions; never 0 inside the loop)
//   Tiles past the end of K are staged from the zero page (uniform instruction counts; they land in slots
//   nobody reads).
template <bool CONV>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmP p)
{
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int STAGE = (BM + BN) * BK * 2;      // 64 KiB
    constexpr int BOFF = BM * BK * 2;              // B tile offset inside a stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- XCD-aware tile mapping + column panels (as gemm_kernel)
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
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
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- staging assignment.  A unit = 128 tile rows = 2 x (8 waves x 8 rows); this thread stages, for every
    // 64-row block it touches, row srow of the block and the LOGICAL chunk whose swizzled slot is lane&7.
    const int srow = wave * 8 + (lane >> 3);                 // 0..63
    const int sc = (lane & 7) ^ ((srow >> 1) & 7);           // every row this thread stages is = srow mod 16
    const _Float16* zsrc = reinterpret_cast<const _Float16*>(g_zero_page);
    const int He = p.ups ? p.H * 2 : p.H, We = p.ups ? p.W * 2 : p.W;

    // A rows: tile row srow + 64*q, q = 0..3   (U1: q = 0,2 ; U4: q = 1,3)
    int a_pix[4], a_ih0[4], a_iw0[4];
    bool a_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + srow + 64 * q;
        a_ok[q] = m < p.M;
        const int mm = a_ok[q] ? m : 0;
        if (CONV) {
            const int ohw = p.OH * p.OW;
            const int img = mm / ohw, rem = mm - img * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_pix[q] = img * p.H * p.W;
            a_ih0[q] = oh * p.stride - p.pad;
            a_iw0[q] = ow * p.stride - p.pad;
        } else {
            a_pix[q] = mm; a_ih0[q] = a_iw0[q] = 0;
        }
    }
    // B rows: unit row u = it*64 + srow -> tile row (u>>5)*64 + (u&31) (+32 for U3)
    int b_row[4];          // index = it + 2*(unit is U3)
    bool b_ok[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int it = x & 1, u = it * 64 + srow;
        const int r = (u >> 5) * 64 + (u & 31) + ((x >> 1) ? 32 : 0);
        b_row[x] = r;
        b_ok[x] = n0 + r < p.N;
    }
    // conv: (kh, kw, cin) of this thread's chunk, one running position per A unit sequence (U1 and U4 are issued
    // for different K tiles in the same phase window)
    int ck_kh[2] = {0, 0}, ck_kw[2] = {0, 0}, ck_cin[2] = {sc * 8, sc * 8};
    if (CONV) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            while (ck_cin[s] >= p.Cin) { ck_cin[s] -= p.Cin; if (++ck_kw[s] == p.KW) { ck_kw[s] = 0; ++ck_kh[s]; } }
    }

    auto issue_A = [&](int kt, int second /* 0: U1, 1: U4 */) {
        unsigned char* stage = smem + (kt & 1) * STAGE;
        const int k = kt * BK + sc * 8;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * 2 + second;
            const _Float16* src = zsrc;
            if (CONV) {
                const int ih = a_ih0[q] + ck_kh[second], iw = a_iw0[q] + ck_kw[second];
                if (a_ok[q] && ck_kh[second] < p.KH && (unsigned)ih < (unsigned)He && (unsigned)iw < (unsigned)We) {
                    const int sh = p.ups ? (ih >> 1) : ih, sw = p.ups ? (iw >> 1) : iw;
                    src = p.A + (long)(a_pix[q] + sh * p.W + sw) * p.lda + ck_cin[second];
                }
            } else {
                if (a_ok[q] && k < p.K) src = p.A + (long)a_pix[q] * p.lda + k;
            }
            unsigned char* dst = stage + (q * 64 + wave * 8) * (BK * 2);       // wave-uniform: 8 rows, lane-linear
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        if (CONV) {
            ck_cin[second] += BK;
            while (ck_cin[second] >= p.Cin) { ck_cin[second] -= p.Cin; if (++ck_kw[second] == p.KW) { ck_kw[second] = 0; ++ck_kh[second]; } }
        }
    };
    auto issue_B = [&](int kt, int second /* 0: U2, 1: U3 */) {
        unsigned char* stage = smem + (kt & 1) * STAGE + BOFF;
        const int k = kt * BK + sc * 8;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int x = it + 2 * second;
            const _Float16* src = (b_ok[x] && k < p.K) ? p.B + (long)(n0 + b_row[x]) * p.ldb + k : zsrc;
            const int r0 = (it * 2 + (wave >> 2)) * 64 + (wave & 3) * 8 + second * 32;   // first of the wave-instruction's 8 rows
            unsigned char* dst = stage + r0 * (BK * 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // ---- fragment addressing (16x16x32: lane = (row l15, k group lg)); rows are = l15 mod 16, so the swizzle
    // term is lane-constant: slot(ks) = ((4*ks + lg) ^ (l15 >> 1)) = c0 ^ (4*ks)
    const int c0 = lg ^ (l15 >> 1);
    const int fa = (wr * 128 + l15) * (BK * 2);                // + (qa*64 + i*16) rows
    const int fb = BOFF + (wc * 64 + l15) * (BK * 2);          // + (qb*32 + j*16) rows
    const int fk0 = c0 << 4, fk1 = (c0 ^ 4) << 4;

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 af[4][2], bf[2][2][2];

    auto read_A = [&](const unsigned char* stage, int qa) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* r = stage + fa + (qa * 64 + i * 16) * (BK * 2);
            af[i][0] = *reinterpret_cast<const f16x8*>(r + fk0);
            af[i][1] = *reinterpret_cast<const f16x8*>(r + fk1);
        }
    };
    auto read_B = [&](const unsigned char* stage, int qb) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* r = stage + fb + (qb * 32 + j * 16) * (BK * 2);
            bf[qb][j][0] = *reinterpret_cast<const f16x8*>(r + fk0);
            bf[qb][j][1] = *reinterpret_cast<const f16x8*>(r + fk1);
        }
    };

    const int nkt = (p.K + BK - 1) / BK;

    // ---- prologue: U1..U4 of tile 0 and U1 of tile 1 in flight; U1(0), U2(0) retired and visible
    issue_A(0, 0); issue_B(0, 0); issue_B(0, 1); issue_A(0, 1); issue_A(1, 0);
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
                    acc[QA][QB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][ks], bf[QB][j][ks], acc[QA][QB][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char* stage = smem + (kt & 1) * STAGE;
        MLSD_PP_PHASE(0, 0, true, true, issue_B(kt + 1, 0), true)
        MLSD_PP_PHASE(0, 1, false, true, issue_B(kt + 1, 1), true)
        MLSD_PP_PHASE(1, 1, true, false, issue_A(kt + 1, 1), false)
        MLSD_PP_PHASE(1, 0, false, false, issue_A(kt + 2, 0), true)
    }
#undef MLSD_PP_PHASE
    if (wr == 0) __builtin_amdgcn_s_barrier();     // rejoin the groups
    wait_vmcnt<0>();                               // the zero-page tail stages still target the ring
    __syncthreads();

    // ---- epilogue.  acc[qa][qb][i][j][e]: row = qa*64 + i*16 + 4*lg + e, col = qb*32 + j*16 + l15 (inside the wave's
    // 128x64).  32 rows x 64 columns at a time through the wave's 8 KiB of LDS, then a lane owns 4 consecutive columns.
    float* const C32 = p.C32 ? p.C32 + (long)blockIdx.y * p.ws_stride : nullptr;
    const bool geglu = p.act == MLSD_ACT_GEGLU;
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const int wrow0 = m0 + wr * 128, wcol0 = n0 + wc * 64;
    auto put_slab = [&](auto s_const) {
        constexpr int s = decltype(s_const)::value;            // 32-row slab 0..3
        constexpr int qa = s >> 1, ib = (s & 1) * 2;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        stg[(ii * 16 + 4 * lg + e) * 64 + qb * 32 + j * 16 + l15] = acc[qa][qb][ib + ii][j][e];
    };
    auto emit_slab = [&](int s) {
        const int mbase = wrow0 + s * 32;
        if (p.vec && !geglu) {
            const int c4 = (lane & 15) * 4;
            const int n = wcol0 + c4;
            float4 bv = make_float4(0, 0, 0, 0);
            if (p.bias && n < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int m = mbase + row;
                float4 v = *reinterpret_cast<const float4*>(stg + row * 64 + c4);
                if (m >= p.M || n >= p.N) continue;
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
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
                if (C32) *reinterpret_cast<float4*>(C32 + (long)m * p.ldc32 + n) = v;
                if (p.C16) {
                    f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                    *reinterpret_cast<f16x4*>(p.C16 + (long)m * p.ldc16 + n) = h;
                }
            }
        } else if (p.vec) {
            // GEGLU: slab columns 0..31 = value, 32..63 = gate (weight rows interleaved in blocks of 32)
            const int c4 = (lane & 7) * 4;
            const int nv = wcol0 + c4, ng = nv + 32;
            const int no = (wcol0 >> 6) * 32 + c4;
            float4 bvv = make_float4(0, 0, 0, 0), bgg = bvv;
            if (p.bias && ng < p.N) { bvv = *reinterpret_cast<const float4*>(p.bias + nv); bgg = *reinterpret_cast<const float4*>(p.bias + ng); }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int m = mbase + row;
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
        } else {
            // scalar fallback (N or a stride not a multiple of 4): lane handles column pairs (lane, lane+... ) row by row
            for (int row = 0; row < 32; ++row) {
                const int m = mbase + row;
                if (m >= p.M) break;
                const float* rbias = p.rowbias ? p.rowbias + (long)(m / p.rows_per_batch) * p.ldrb : nullptr;
                if (!geglu) {
                    const int n = wcol0 + lane;
                    if (n < p.N) {
                        float v = stg[row * 64 + lane];
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
                } else if (lane < 32) {
                    const int nv = wcol0 + lane, ng = nv + 32;
                    if (ng < p.N) {
                        float v = stg[row * 64 + lane], g = stg[row * 64 + 32 + lane];
                        if (p.bias) { v += p.bias[nv]; g += p.bias[ng]; }
                        v = v * gelu_tanh_f(g);
                        const int no = (wcol0 >> 6) * 32 + lane;
                        if (p.resid) v += p.resid[(long)m * p.ldr + no];
                        if (C32) C32[(long)m * p.ldc32 + no] = v;
                        if (p.C16) p.C16[(long)m * p.ldc16 + no] = (_Float16)v;
                    }
                }
            }
        }
    };
    put_slab(std::integral_constant<int, 0>{}); __builtin_amdgcn_wave_barrier(); emit_slab(0); __builtin_amdgcn_wave_barrier();
    put_slab(std::integral_constant<int, 1>{}); __builtin_amdgcn_wave_barrier(); emit_slab(1); __builtin_amdgcn_wave_barrier();
    put_slab(std::integral_constant<int, 2>{}); __builtin_amdgcn_wave_barrier(); emit_slab(2); __builtin_amdgcn_wave_barrier();
    put_slab(std::integral_constant<int, 3>{}); __builtin_amdgcn_wave_barrier(); emit_slab(3); __builtin_amdgcn_wave_barrier();
}
