// Persistent "ping-pong" GEMM tiles (256x256x64 and 128x320x64) for gfx950 — included by gemm_conv.hip (inside
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

template <int BM, int BN>
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
    return PPTile{bm * BM, bn * BN};
}

// tile `bid` of the panel order, WITHOUT the XCD remap (stream-K: the block numbering is XCD-contiguous already)
template <int BM, int BN>
__device__ __forceinline__ PPTile pp_tile_plain(const GemmP& p, int bid)
{
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
    return PPTile{bm * BM, bn * BN};
}

// STREAM-K (template parameter SK).  With T output tiles on G = 256 persistent blocks a launch takes
// ceil(T / G) whole tile times: 8192 x 1280 on 256 x 256 tiles is 160 tiles = 62 % of the chip, which is why those outputs ran
// on the less efficient 128 x 320 tile (256 tiles).  Here the T * (K / 64) K-tile UNITS of the launch are dealt out evenly instead:
// block w (numbered XCD-contiguously: the blocks that share a tile share an L2) takes units [w L, (w + 1) L), L = ceil(units / G),
// in tile-major order -- its stream starts in the middle of a tile, may contain whole tiles, and ends in the middle of one:
//   * a segment that STARTS inside a tile (only a block's first one) is a contribution: at its end the block writes its raw
//     accumulators to its slab of the workspace (write-through stores), drains them, and raises its flag;
//   * a segment that starts AT a tile's first K tile makes the block the tile's OWNER: if its stream ends before the tile's last K
//     tile it waits, in block order, for the flags of the blocks holding the rest (they never wait for anybody: no cycle, no
//     residency assumption), adds their slabs in that fixed order (bit-repeatable) and runs the normal epilogue; each flag has
//     exactly one consumer, which clears it for the next launch.
// Everything else -- the staging sequences, the phases, the counted waits -- is the persistent kernel's: a stream of K tiles that
// crosses tile seams was already its model.  (Hand-off: cdna_hip_programming.md section 5 "In-launch split-K reduction", the
// write-through form.)
// Geometry (template): the block tile is BM x BN, waves 2 x 4, a wave owns WM x WN = (BM/2) x (BN/4):
//   256 x 256: wave 128 x 64, quadrants 64 x 32 (CB0 = CB1 = 2 column blocks of 16)   -- the description above
//   128 x 320: wave  64 x 80, quadrants 32 x 48 | 32 x 32 (CB0 = 3, CB1 = 2): N = 1280 / 640 / 320 outputs in whole tile
//              columns and 8192 x 1280 = exactly 256 tiles; residual rows are fetched in ONE batch at the start of the
//              epilogue (80 spare registers) so their latency is paid once per output tile, not once per row block
// Units are cut the same way (U1/U4 = the a0/a1 rows of both wave rows, U2/U3 = the b0/b1 rows of all four wave
// columns); a unit of R rows is R/64 LDS-DMA instructions per thread, and the counted waits follow:
//   P4: NU3 + NU4 + NU1 younger instructions,  P1: NU4 + NU1 + NU2,  P2: NU1 + NU2 + NU3     (6/6/6 and 4/5/6)
// Linear only (plain row-major A), M a multiple of BM/2, N of BN/4, K a multiple of 64 and >= 192 (3 K tiles: U1, two
// tiles ahead, must enter output tile ti+1 while the MFMAs are in tile ti): the per-phase staging code is LDS-DMA
// instructions on running row pointers and nothing else (out-of-range rows are CLAMPED to the last valid row instead of
// zero-filled: their products land in output rows/columns that are never stored).
// EPI selects the epilogue the kernel is BUILT with (launch_pp picks it from the arguments): 0 = generic (any activation,
// per-row bias, any output combination: decided per element at run time), 1 = fp16 out, 2 = fp32 out, 3 = fp32 out + fp32
// residual, 4 = GEGLU fp16 out (256-wide tile); 1..4: no activation / per-row bias, straight-line code (see epi_fast).
enum { PP_EPI_GENERIC = 0, PP_EPI_F16 = 1, PP_EPI_F32 = 2, PP_EPI_F32_RES = 3, PP_EPI_GEGLU16 = 4, PP_EPI_F32_STATS = 5, PP_EPI_F32_RES_STATS = 6,
       PP_EPI_F32_LN = 7, PP_EPI_F32_RES_LN = 8, PP_EPI_XATTN = 9 };
// XATTN (round 6, 128 x 320 tile = 128 query rows x 5 heads of 64, one output tile per block): the launch is the q projection of a cross attention and ENDS with that
// attention (mlsd_gemm_args.xa_*).  Nothing of q reaches HBM.  After the K loop (LDS map: K [77][656 B] at 0, q [128][656 B] at 50 KB; V^T [320][208 B] later over q):
//   1. the tail stages of the ring are drained (they target the LDS this epilogue re-uses), the image's K rows of the tile's 320 columns go global -> LDS by LDS-DMA;
//   2. every wave rounds its 64 x 80 accumulators to fp16 (the rounding point of the unfused launch's C16) and writes them into the q image (80-column wave tiles
//      straddle the 64-column heads: q changes hands through LDS, 80 KB written and read once);
//   3. wave w takes query rows 16 w .. 16 w + 15 of all 5 heads: its 10 q fragments (B operands) go to registers, the q image is dead, V^T is DMA'd over it and lands
//      while the scores are computed;
//   4. per head: S^T[key][row] = K_h . q_h^T (5 key tiles x 2 MFMAs), mask keys >= Tk, softmax over the key axis (in lane + two cross-group exchanges), P -> fp16 in the
//      k order the accumulator tiles give (elements 0-3: key 32 ks + 4 lg + e of tile 2 ks, 4-7: key 32 ks + 16 + 4 lg + e of tile 2 ks + 1);
//   5. per head: O^T[d][row] = V_h^T . P (4 d tiles x 3 MFMAs; the V^T fragments are read in the same permuted key order), O / l -> fp16, 16-byte stores.
// Same rounding points as the unfused pair (fp16 q, K, V, P; fp32 scores, softmax statistics and accumulation); the fp32 summation orders differ from attn_tk96's.
// *_LN (round 3, 128 x 320 tile, single-round linear launches): the launch ENDS with the LayerNorm of its fp32 output instead of being followed by one.  The 4 (N = 1280)
// column tiles of a row block run at the same time on CUs of one XCD; each computes, per row, the mean and the centred sum of squares of its 320 columns (4 lanes by shuffles,
// 4 wave columns through 4 KB of LDS, Chan's pairwise combination: no cancellation) and publishes them as self-tagged records (round 6: {mean, tag, m2, tag}, write-through;
// the tag is the launch's epoch + 1, the scratch is the launch's own: epi_ln step 4).  ONE wave per (block, wave row) loads the partners' records until their tags match
// (every wave polling starved the arrivals: 30 us per launch), the others wait at a barrier and take the values from LDS; then every wave combines them in tile order
// (bit-repeatable), normalises the values it still holds in its accumulators and writes the fp16 rows.  No ticket, no arrival / departure counters, nothing to reset (rounds 3-5
// had all three: three serial round trips per launch).  Bounded polling; a give-up raises a sticky word (ln_cnt[8191]) that the host reads with the results.  Measured
// (tools/ln_fold_bench.py): 8192x1280x1280 GEMM 40.3 + LayerNorm 12.0 = 53.3 us as two launches, 50.9 us as one with the counters (46.6 without any exchange).
// *_STATS: additionally the column sums / sums of squares of the wave's rows (GemmP::colstats) for a consuming GroupNorm
// NPH = 2: TWO phases per K tile instead of four (round 3).  In-kernel stamps of the four-phase loop on the 128 x 320 tile: 2075 clocks
// per K tile against 1280 of matrix work, the same with the staging switched off (profiles/r2_gemm_trace_*: the loop is not fill-bound
// on this tile) -- each of its 8 barrier-to-barrier sections carries only 8 or 12 MFMAs (128 / 192 clocks) against ~100 clocks of
// fixed cost per section (barrier hand-over, priority switch, issue of the next section's reads).  With two phases a section is 20
// MFMAs (320 clocks):   A: all B fragments + the a0 rows, MFMAs of acc[0];   B: the a1 rows (B fragments stay in registers), acc[1].
// Staging, every unit ONE K tile ahead:   slot A: U1, U2, U3 of tile s+1, counted wait retires U4(s);   slot B: U4 of tile s+1,
// counted wait retires U1..U3(s+1).  RAW: a unit is read one slot after the wait that retires it, by then every wave of both groups
// has passed that wait and a barrier.  WAR: a stage is refilled two or more barriers after its last fragment read was consumed.
// SCH = 1 (four phases, 128 x 320 tile): the SAME units on a re-balanced issue schedule.  Timing-only builds of the loop (MLSD_PP_DBG,
// profiles/r3_gemm_loop_ablation.txt) price a fragment read at ~15 clocks and an LDS-DMA instruction at ~45 of the wave that issues it,
// and show the loop bound by its load slots, not by the matrix pipe: a group's load slot runs beside the other group's MFMA section, and the
// schedule above puts 10 reads + 3 DMA (~285 clocks) in P1's slot against a 192-clock section while P4's slot is nearly empty.  Here:
//       P1: U4(s+1)     P2: U3(s+1)     P3: -     P4: U1(s+2), U2(s+2)         (U2 now two K tiles ahead, like U1: its rows are read in P1 only)
//   counted waits (what may stay in flight):   P4: NU1+NU2+NU3+NU4 (retires U1, U2 of the next tile)    P1: NU1+NU2+NU4 (retires U4, U3 of this tile)
//   RAW: P1 reads U1/U2 retired in the previous P4; P2 reads U3 retired in P1; P3 reads U4, older than U3, retired with it.  WAR: every
//   unit is restaged >= 3 phases after the phase that read its rows.
template <int BM, int BN, int CB0, int CB1, bool RESBATCH, int CONV, int EPI, bool SK = false, int NPH = 4, int SCH = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmP p)
{
    constexpr int BK = 64, RB = BK * 2;            // bytes per tile row
    constexpr int WM = BM / 2, WN = BN / 4;
    constexpr int RA = WM / 32;                    // 16-row blocks per A half (a0 / a1)
    constexpr int NCB = CB0 + CB1;                 // 16-column blocks per wave
    static_assert(NCB * 16 == WN && RA * 32 == WM, "wave tile");
    constexpr int STAGE = (BM + BN) * RB;
    constexpr int BOFF = BM * RB;                  // B tile offset inside a stage
    constexpr int NU1 = BM / 128, NU2 = CB0, NU3 = CB1;      // LDS-DMA instructions per thread per unit (NU4 = NU1)
    constexpr int W4 = NU3 + 2 * NU1, W1 = 2 * NU1 + NU2, W2 = NU1 + NU2 + NU3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations (M0) stay on the SALU
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nblk = p.nbm * p.nbn;
    const int G = gridDim.x;
    const int nkt = p.K / BK;
    int ntile = (nblk - (int)blockIdx.x + G - 1) / G;            // output tiles of this block (>= 1)
    int S = ntile * nkt;                                         // K tiles in this block's stream
    int kt0 = 0, tfirst = 0, skw = 0;                            // stream-K: first K tile inside the first tile, first tile, block number
    if constexpr (SK) {
        const int q = G >> 3, r = G & 7, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        skw = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        const long W = (long)nblk * nkt, u0 = (long)skw * p.sk_L;
        if (u0 >= W) return;                                     // (the whole block leaves before any barrier)
        const long u1 = min(u0 + p.sk_L, W);
        S = (int)(u1 - u0); tfirst = (int)(u0 / nkt); kt0 = (int)(u0 - (long)tfirst * nkt);
        ntile = (int)((u1 - 1) / nkt) - tfirst + 1;
    }
    auto tile_at = [&](int ti) __attribute__((always_inline)) {   // coordinates of this block's ti-th output tile
        if constexpr (SK) return pp_tile_plain<BM, BN>(p, min(tfirst + ti, nblk - 1));
        else return pp_tile_coords<BM, BN>(p, (int)blockIdx.x + ti * G);
    };
    int kenter = kt0;                                            // K tile the staging sequences start at when they enter a tile (kt0 once, then 0)

    // ---- staging.  A wave-instruction fills 8 consecutive tile rows (lane l -> row l>>3, slot l&7); instruction
    // g = it*8 + wave of a unit covers the g-th group of 8 rows of the unit's row list.  The thread fetches the LOGICAL
    // chunk whose swizzled slot is lane&7 in ITS row (baked into the running pointer).
    auto a_row0 = [&](int it, int second) __attribute__((always_inline)) {        // U1 (second = 0) / U4 (1)
        const int g = it * 8 + wave;
        return (g / (WM / 16)) * WM + (g % (WM / 16)) * 8 + second * (WM / 2);
    };
    auto b_row0 = [&](int it, int second) __attribute__((always_inline)) {        // U2 (second = 0) / U3 (1)
        const int per = (second ? CB1 : CB0) * 2;
        const int g = it * 8 + wave;
        return (g / per) * WN + (g % per) * 8 + second * (CB0 * 16);
    };
    auto chunk_of = [&](int row) __attribute__((always_inline)) { return ((lane & 7) ^ ((row >> 1) & 7)) * 8; };

    // One iterator per unit sequence: U1 runs two K tiles ahead of the MFMAs, U2..U4 one.  Each holds running row
    // pointers (this thread's chunk of the current K tile) for the output tile it is staging for; all of them enter
    // output tile ti+1 while the MFMAs are still in tile ti, so "the next tile" is one shared coordinate pair.
    // CONV (3x3 / 1x1, stride 1 / 2, no upsample, Cin % 64 == 0): a K tile lies inside ONE filter tap, so the tap and the
    // channel offset are block-uniform (scalar); a row keeps the address of its output pixel's top-left input pixel
    // and a 9-bit mask of the taps that fall inside the image; padded taps read the zero page.
    // CONV == 2 (round 5): the source is read through a nearest-2x upsample (ggml_upscale + conv, src/mlblock_nn.c:122; stride 1): tap (kh, kw) of output pixel (oh, ow) is
    // source pixel ((oh - pad + kh) >> 1, (ow - pad + kw) >> 1), so the tap offset depends on the parity of the row's pixel: the row keeps its image base and its
    // (oh - pad, ow - pad) pair, and the source address is formed per row and K tile (a handful of VALU operations per LDS-DMA instruction) instead of once per tile.
    constexpr bool UPS = CONV == 2;
    struct SeqA { int kt, par; const _Float16* ptr[NU1]; int mask[CONV ? NU1 : 1]; int pos[UPS ? NU1 : 1]; int kh, kw, cin; };
    struct SeqB2 { int kt, par; const _Float16* ptr[NU2]; };
    struct SeqB3 { int kt, par; const _Float16* ptr[NU3]; };
    SeqA sa[2];    // [0] = U1, [1] = U4
    SeqB2 sb2;     // U2
    SeqB3 sb3;     // U3
    PPTile tnext = tile_at(0);                                   // coordinates the sequences use at their next tile entry
    PPTile tcur = tnext;                                         // tile of the MFMAs / next epilogue
    const _Float16* zsrc = reinterpret_cast<const _Float16*>(g_zero_page);

    auto enter_A = [&](SeqA& s, int second) __attribute__((always_inline)) {
        s.kt = kenter; s.kh = 0; s.kw = 0; s.cin = 0;
        if constexpr (CONV) {
            if (kenter) {
#ifdef MLSD_GEMM_EXPERIMENTS      /* slab-ordered K (round-5 experiment, timing only: mlsd_gemm_set_korder) */
                if (p.korder) { const int kk = p.KH * p.KW, slab = kenter / kk, tap = kenter - slab * kk; s.cin = slab * BK; s.kh = tap / p.KW; s.kw = tap - s.kh * p.KW; }
                else
#endif
                { const int k0 = kenter * BK, tap = k0 / p.Cin; s.cin = k0 - tap * p.Cin; s.kh = tap / p.KW; s.kw = tap - s.kh * p.KW; }
            }
        }
#pragma unroll
        for (int it = 0; it < NU1; ++it) {
            const int r = a_row0(it, second) + (lane >> 3);
            const int m = min(tnext.m0 + r, p.M - 1);
            if constexpr (CONV) {
                const int ohw = p.OH * p.OW;
                const int img = m / ohw, rem = m - img * ohw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
                const int He = UPS ? 2 * p.H : p.H, We = UPS ? 2 * p.W : p.W;
                int mk = 0;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw)
                        if ((unsigned)(ih0 + kh) < (unsigned)He && (unsigned)(iw0 + kw) < (unsigned)We) mk |= 1 << (kh * p.KW + kw);
                s.mask[it] = mk;
                if constexpr (UPS) {
                    s.pos[it] = ((ih0 + 16) << 16) | (iw0 + 16);                  // (pad <= 16: both halves stay non-negative)
                    s.ptr[it] = p.A + (long)img * p.H * p.W * p.lda + chunk_of(r);
                } else
                s.ptr[it] = p.A + ((long)img * p.H * p.W + (long)ih0 * p.W + iw0) * p.lda + chunk_of(r);
            } else {
                s.ptr[it] = p.A + (long)m * p.lda + chunk_of(r) + kenter * BK;
            }
        }
    };
    auto enter_B2 = [&](SeqB2& s) __attribute__((always_inline)) {
        s.kt = kenter;
#pragma unroll
        for (int it = 0; it < NU2; ++it) {
            const int r = b_row0(it, 0) + (lane >> 3);
            s.ptr[it] = p.B + (long)min(tnext.n0 + r, p.N - 1) * p.ldb + chunk_of(r) + kenter * BK;
        }
    };
    auto enter_B3 = [&](SeqB3& s) __attribute__((always_inline)) {
        s.kt = kenter;
#pragma unroll
        for (int it = 0; it < NU3; ++it) {
            const int r = b_row0(it, 1) + (lane >> 3);
            s.ptr[it] = p.B + (long)min(tnext.n0 + r, p.N - 1) * p.ldb + chunk_of(r) + kenter * BK;
        }
    };
    auto issue_A = [&](SeqA& s, int second) __attribute__((always_inline)) {
        unsigned char* stage = smem + s.par * STAGE;
        const long toff = CONV ? ((long)s.kh * p.W + s.kw) * p.lda + s.cin : 0;      // scalar: tap + channel offset of this K tile
        const int tbit = CONV ? 1 << (s.kh * p.KW + s.kw) : 0;
#pragma unroll
        for (int it = 0; it < NU1; ++it) {
            unsigned char* dst = stage + a_row0(it, second) * RB;                  // wave-uniform: 8 rows, lane-linear
            const _Float16* src = s.ptr[it];
            if constexpr (UPS) {
                const int ih = (s.pos[it] >> 16) - 16 + s.kh, iw = (s.pos[it] & 0xffff) - 16 + s.kw;
                src = (s.mask[it] & tbit) ? s.ptr[it] + ((long)(ih >> 1) * p.W + (iw >> 1)) * p.lda + s.cin : zsrc;
            } else
            if constexpr (CONV) src = (s.mask[it] & tbit) ? s.ptr[it] + toff : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            if constexpr (!CONV) s.ptr[it] += BK;
        }
        s.par ^= 1;
        if constexpr (CONV) {
#ifdef MLSD_GEMM_EXPERIMENTS
            if (p.korder) { if (++s.kw == p.KW) { s.kw = 0; if (++s.kh == p.KH) { s.kh = 0; s.cin += BK; } } }      // slab order: the taps of this 64-channel slab first
            else
#endif
            { s.cin += BK; if (s.cin == p.Cin) { s.cin = 0; if (++s.kw == p.KW) { s.kw = 0; ++s.kh; } } }
        }
        if (++s.kt == nkt) enter_A(s, second);
    };
    auto issue_B2 = [&]() __attribute__((always_inline)) {
        unsigned char* stage = smem + sb2.par * STAGE + BOFF;
#pragma unroll
        for (int it = 0; it < NU2; ++it) {
            unsigned char* dst = stage + b_row0(it, 0) * RB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb2.ptr[it],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            sb2.ptr[it] += BK;
        }
        sb2.par ^= 1;
        if (++sb2.kt == nkt) enter_B2(sb2);
    };
    auto issue_B3 = [&]() __attribute__((always_inline)) {
        unsigned char* stage = smem + sb3.par * STAGE + BOFF;
#pragma unroll
        for (int it = 0; it < NU3; ++it) {
            unsigned char* dst = stage + b_row0(it, 1) * RB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb3.ptr[it],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            sb3.ptr[it] += BK;
        }
        sb3.par ^= 1;
        if (++sb3.kt == nkt) enter_B3(sb3);
    };

    // ---- fragment addressing (16x16x32: lane = (row l15, k group lg)); rows are = l15 mod 16, so the swizzle
    // term is lane-constant: slot(ks) = ((4*ks + lg) ^ (l15 >> 1)) = c0 ^ (4*ks)
    const int c0 = lg ^ (l15 >> 1);
    const int fa = (wr * WM + l15) * RB;                       // + (qa*WM/2 + i*16) rows
    const int fb = BOFF + (wc * WN + l15) * RB;                // + cbi*16 rows
    const int fk0 = c0 << 4, fk1 = (c0 ^ 4) << 4;

    f32x4 acc[2][RA][NCB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < RA; ++i)
#pragma unroll
                for (int c = 0; c < NCB; ++c) acc[a][i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    f16x8 af[RA][2] = {}, bf[NCB][2] = {};

    auto read_A = [&](const unsigned char* stage, int qa) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const unsigned char* r = stage + fa + (qa * (WM / 2) + i * 16) * RB;
            af[i][0] = *reinterpret_cast<const f16x8*>(r + fk0);
            af[i][1] = *reinterpret_cast<const f16x8*>(r + fk1);
        }
    };
    auto read_B = [&](const unsigned char* stage, int qb) __attribute__((always_inline)) {
#pragma unroll
        for (int c = (qb ? CB0 : 0); c < (qb ? NCB : CB0); ++c) {
            const unsigned char* r = stage + fb + c * 16 * RB;
            bf[c][0] = *reinterpret_cast<const f16x8*>(r + fk0);
            bf[c][1] = *reinterpret_cast<const f16x8*>(r + fk1);
        }
    };

    // ---- epilogue of one output tile, straight from the accumulators.  The MFMAs compute the TRANSPOSED product
    // (B fragment as the first operand), so acc[qa][i][c][e] = C[row qa*WM/2 + i*16 + l15][col c*16 + 4*lg + e] inside the
    // wave's WM x WN: a lane holds 4 CONSECUTIVE COLUMNS of one row -> 16-byte fp32 / 8-byte fp16 accesses with no LDS
    // transposition (the LDS pass of the other tiles costs ~9 instructions per element; here ~1), and GEGLU's value
    // (column blocks 0,1) and gate (blocks 2,3 of the 256-wide tile) of one output element sit in the same lane.
    // The epilogue is straight-line code instantiated per accumulator tile, executed once per output tile: it must stay
    // SMALL (a first version with a per-element activation switch and scalar fallbacks was 100 KB of code and cost
    // 12 us per tile in instruction fetch).  Hence: whole wave blocks and 16-byte-aligned layouts only (the launcher sends
    // anything else to the LDS-transposing tiles) and ONE activation formula  y = x * sigmoid(x * (c1 + c3 x^2))  for SiLU
    // (1, 0), quick-GELU (1.702, 0) and tanh-GELU (2c, 2c*0.044715), or  y = max(x, lo)  for none (lo = -inf) / ReLU (0).
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
    f32x4 cb[NCB];                                 // bias (+ row bias) of the lane's column groups, loaded once per output tile
    f32x4 rpre[RESBATCH ? 2 * RA * NCB : 1];       // RESBATCH: the whole residual tile of the lane, fetched in one batch
    auto epi_rows = [&](auto QA, auto I, int wrow0, int wcol0) __attribute__((always_inline)) {
        constexpr int qa = decltype(QA)::value, i = decltype(I)::value;
        const int m = wrow0 + qa * (WM / 2) + i * 16 + l15;
        const float bm = p.biasm ? p.biasm[m] : 0.f;
        if (!geglu) {
            const long col = wcol0 + 4 * lg;
            const float* rrow = p.resid ? p.resid + (long)m * p.ldr + col : nullptr;
            float* c32 = p.C32 ? p.C32 + (long)m * p.ldc32 + col : nullptr;
            _Float16* c16 = p.C16 ? p.C16 + (long)m * p.ldc16 + col : nullptr;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                f32x4 v = acc[qa][i][c] + cb[c];
                v += bm;
                f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                if constexpr (RESBATCH) { if (p.resid) rs = rpre[(qa * RA + i) * NCB + c]; }
                else { if (rrow) rs = *reinterpret_cast<const f32x4*>(rrow + c * 16); }
                if (p.act_post) v += rs;
                v = act4(v);
                if (!p.act_post) v += rs;
                if (c32) *reinterpret_cast<f32x4*>(c32 + c * 16) = v;
                if (c16) {
                    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4*>(c16 + c * 16) = h;
                }
            }
        } else if constexpr (CB0 == 2 && CB1 == 2) {
            const long col = (wcol0 >> 6) * 32 + 4 * lg;     // value | gate column blocks of 32 are interleaved
            const float* rrow = p.resid ? p.resid + (long)m * p.ldr + col : nullptr;
            float* c32 = p.C32 ? p.C32 + (long)m * p.ldc32 + col : nullptr;
            _Float16* c16 = p.C16 ? p.C16 + (long)m * p.ldc16 + col : nullptr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 a4 = acc[qa][i][j] + cb[j], g4 = acc[qa][i][2 + j] + cb[2 + j];
                f32x4 v;
                v[0] = a4[0] * gelu_tanh_f(g4[0]); v[1] = a4[1] * gelu_tanh_f(g4[1]);
                v[2] = a4[2] * gelu_tanh_f(g4[2]); v[3] = a4[3] * gelu_tanh_f(g4[3]);
                if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + j * 16);
                if (c32) *reinterpret_cast<f32x4*>(c32 + j * 16) = v;
                if (c16) {
                    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4*>(c16 + j * 16) = h;
                }
            }
        }
    };
    // FAST VARIANTS.  The generic rows above decide per element what to do (null checks on per-lane pointers, the activation
    // kind, act-before/after-residual): ~30 instructions and 5 taken branches per 4 outputs, measured at 7 us per 256x256
    // tile WITHOUT its stores (tools/gemm_ksweep.py "no-stores") -- a quarter of a K = 1280 tile.  Everything the UNet
    // launches in bulk has no activation in the GEMM (SiLU/GELU live in the norm kernels or the GEGLU path) and no per-row
    // bias, so those launches use a kernel BUILT with one straight-line body (template parameter EPI; a run-time choice
    // between bodies inside one kernel cost 20-70 spilled registers in the K loop): a 4-output group is 1-2 packed adds,
    // the conversion and the store.
    // Residual rows are fetched PF row blocks ahead of their use (all of them on the 128-row tile: 80 registers that the
    // main loop's fragments no longer need; two blocks = 32 registers on the 256-row tile).
    auto epi_fast = [&](auto S32_, auto S16_, auto RES_, auto ST_, int wrow0, int wcol0) __attribute__((always_inline)) {
        constexpr bool S32 = decltype(S32_)::value, S16 = decltype(S16_)::value, RES = decltype(RES_)::value, ST = decltype(ST_)::value;
        constexpr int NR = 2 * RA;                               // 16-row blocks of the wave's slab: r = qa * RA + i
        constexpr int PF = !ST ? (BM == 128 ? NR : 2) : (BM == 128 ? NR : 1);  // (the statistics need 2 * NCB * 4 registers of their own)
        // ST: per column, sum and sum of squares over the wave's WM rows (a lane accumulates its 2*RA rows, the 16 lanes of a
        // DPP row are the 16 rows of a block) -> one deterministic partial per (tile row, wave row, column): GemmP::colstats
        f32x4 cs[ST ? NCB : 1], cq[ST ? NCB : 1];
        if constexpr (ST) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) { cs[c] = f32x4{0.f, 0.f, 0.f, 0.f}; cq[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        f32x4 rr[RES ? NR : 1][RES ? NCB : 1];
        const float* rbase = RES ? p.resid + (long)(wrow0 + l15) * p.ldr + wcol0 + 4 * lg : nullptr;
        auto row_of = [&](int r) __attribute__((always_inline)) { return (r / RA) * (WM / 2) + (r % RA) * 16; };
        auto fetch = [&](int r) __attribute__((always_inline)) {
            if constexpr (RES) {
                const float* rp = rbase + (long)row_of(r) * p.ldr;
#pragma unroll
                for (int c = 0; c < NCB; ++c) rr[r][c] = *reinterpret_cast<const f32x4*>(rp + c * 16);
            }
        };
#pragma unroll
        for (int r = 0; r < PF && r < NR; ++r) fetch(r);
        float* c32b = S32 ? p.C32 + (long)(wrow0 + l15) * p.ldc32 + wcol0 + 4 * lg : nullptr;
        _Float16* c16b = S16 ? p.C16 + (long)(wrow0 + l15) * p.ldc16 + wcol0 + 4 * lg : nullptr;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + PF < NR) fetch(r + PF);
            const int qa = r / RA, i = r % RA;
            if constexpr (S16 && !S32 && !RES && !ST) {
                // fp16 only: the row's 16 columns of a block sit in 4 lanes (lg) as 8-byte pieces.  The tail is store-ISSUE bound
                // (~34 clocks per wave store, gemm_trace; tools/store_ab.py measured 8192x3840x1280 84.2 -> 75.9 us), so pairs of column blocks exchange halves between lane groups 0<->1 and
                // 2<->3 (v_permlane16_swap) and every lane stores 16 contiguous bytes: half the store instructions, same bytes
                // (cdna_hip_programming.md T21).  lg 0/2 end up with block c, lg 1/3 with block c+1, columns 8 (lg >> 1) .. +7.
                _Float16* rowp = p.C16 + (long)(wrow0 + l15 + row_of(r)) * p.ldc16 + wcol0 + 8 * (lg >> 1) + 16 * (lg & 1);
#pragma unroll
                for (int c = 0; c + 1 < NCB; c += 2) {
                    const f32x4 v0 = acc[qa][i][c] + cb[c], v1 = acc[qa][i][c + 1] + cb[c + 1];
                    const f16x4 h0 = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3]};
                    const f16x4 h1 = {(_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                    u32x2 a = __builtin_bit_cast(u32x2, h0), b = __builtin_bit_cast(u32x2, h1);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    *reinterpret_cast<u32x4*>(rowp + c * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
                if constexpr (NCB & 1) {        // odd block count (128x320 tile): the last block keeps its 8-byte pieces
                    const f32x4 v = acc[qa][i][NCB - 1] + cb[NCB - 1];
                    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4*>(c16b + (long)row_of(r) * p.ldc16 + (NCB - 1) * 16) = h;
                }
            } else {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                f32x4 v = acc[qa][i][c] + cb[c];
                if constexpr (RES) v += rr[r][c];
                if constexpr (ST) { cs[c] += v; cq[c] += v * v; }
                if constexpr (S32) *reinterpret_cast<f32x4*>(c32b + (long)row_of(r) * p.ldc32 + c * 16) = v;
                if constexpr (S16) {
                    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4*>(c16b + (long)row_of(r) * p.ldc16 + c * 16) = h;
                }
            }
            }
        }
        if constexpr (ST) {
            auto row16_sum = [&](float v) __attribute__((always_inline)) {       // butterfly over the 16 lanes of a DPP row (xor 1, 2, then mirrors)
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
                return v;
            };
            const long rb = (long)(tcur.m0 / BM) * 2 + wr;
            float* st = p.colstats + rb * 2 * p.N + wcol0 + 4 * lg;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { cs[c][e] = row16_sum(cs[c][e]); cq[c][e] = row16_sum(cq[c][e]); }
                if (l15 == 0) {
                    *reinterpret_cast<f32x4*>(st + c * 16) = cs[c];
                    *reinterpret_cast<f32x4*>(st + p.N + c * 16) = cq[c];
                }
            }
        }
    };
    // GEGLU with fp16 output only (the UNet's feed-forward projection): value (column blocks 0,1) and gate (2,3) of an
    // output element sit in the same lane; straight-line as above
    auto epi_fast_geglu = [&](int wrow0, int wcol0) __attribute__((always_inline)) {
        if constexpr (CB0 == 2 && CB1 == 2) {
            _Float16* c16b = p.C16 + (long)(wrow0 + l15) * p.ldc16 + (wcol0 >> 6) * 32 + 4 * lg;
#pragma unroll
            for (int r = 0; r < 2 * RA; ++r) {
                const int qa = r / RA, i = r % RA;
                _Float16* cp = c16b + (long)(qa * (WM / 2) + i * 16) * p.ldc16;
                u32x2 hh[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 a4 = acc[qa][i][j] + cb[j], g4 = acc[qa][i][2 + j] + cb[2 + j];
                    const f16x4 h = {(_Float16)(a4[0] * gelu_tanh_f(g4[0])), (_Float16)(a4[1] * gelu_tanh_f(g4[1])),
                                     (_Float16)(a4[2] * gelu_tanh_f(g4[2])), (_Float16)(a4[3] * gelu_tanh_f(g4[3]))};
                    hh[j] = __builtin_bit_cast(u32x2, h);
                }
                // the two 16-column blocks of the row as ONE 16-byte store per lane (see epi_fast)
                const auto r0 = __builtin_amdgcn_permlane16_swap(hh[0][0], hh[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(hh[0][1], hh[1][1], false, false);
                *reinterpret_cast<u32x4*>(cp - 4 * lg + 8 * (lg >> 1) + 16 * (lg & 1)) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
        }
    };
    // ---- *_LN epilogue (see the enum): fp32 output (+ residual) AND the LayerNorm of the finished rows as fp16
    auto epi_ln = [&](auto RES_, int wrow0, int wcol0) __attribute__((always_inline)) {
        if constexpr (BM == 128 && BN == 320 && !CONV && !SK) {
            constexpr bool RES = decltype(RES_)::value;
            constexpr int NR = 2 * RA;                              // 16-row blocks of the wave's 64 rows
            // the tag of the records this tile wrote in the PREVIOUS launch of this op (0 = never): this launch's carry that + 1.  Every wave reads the tag of the first row IT
            // publishes (step 4), several barriers before it does: nobody else writes that record, and the row block's partner tiles left the previous launch with the same tag
            // (gemm_tt.hip has the story of the per-launch epoch word this replaces)
            const int rbk = tcur.m0 / BM, bnk = tcur.n0 / BN, nbn = p.nbn;
            const unsigned tag = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.ln_ws) + (((long)rbk * nbn + bnk) * BM + wr * 64) * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            auto row_of = [&](int r) __attribute__((always_inline)) { return (r / RA) * (WM / 2) + (r % RA) * 16; };
            // 1. v = acc + bias (+ residual) -> C32, kept in the accumulators
            f32x4 rr[RES ? NR : 1][RES ? NCB : 1];
            if constexpr (RES) {
                const float* rbase = p.resid + (long)(wrow0 + l15) * p.ldr + wcol0 + 4 * lg;
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int c = 0; c < NCB; ++c) rr[r][c] = *reinterpret_cast<const f32x4*>(rbase + (long)row_of(r) * p.ldr + c * 16);
            }
            float* c32b = p.C32 + (long)(wrow0 + l15) * p.ldc32 + wcol0 + 4 * lg;
            float mean_w[NR], m2_w[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int qa = r / RA, i = r % RA;
                float s1 = 0.f;
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    f32x4 v = acc[qa][i][c] + cb[c];
                    if constexpr (RES) v += rr[r][c];
                    acc[qa][i][c] = v;                              // (stored to C32 after the records are on their way: the publish must not wait behind the tile's 160 KB of stores)
                    s1 += (v[0] + v[1]) + (v[2] + v[3]);
                }
                // 2. the row's 80 columns of this wave sit in the 4 lanes l15 + 16 lg: mean, then the centred sum of squares
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                const float mu = s1 * (1.0f / WN);
                float q = 0.f;
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const f32x4 d = acc[qa][i][c] - mu;
                    q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                }
                q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
                mean_w[r] = mu; m2_w[r] = q;
            }
            // 3. the 4 wave columns of a wave row through LDS (beyond the ring: its tail stages may still be landing), combined in wave-column order
            f32x2* red = reinterpret_cast<f32x2*>(smem + 2 * STAGE);          // [wave row][64 rows][4 wave columns]
            if (lg == 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) red[(wr * 64 + row_of(r) + l15) * 4 + wc] = f32x2{mean_w[r], m2_w[r]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the ds_writes above have landed before the barrier lets the readers through (gemm_tt.hip: the same
                                                                             // exchange with two blocks on the CU read a stale pair once in ~100 launches when the wait stood BEHIND the barrier)
            __builtin_amdgcn_s_barrier();                                    // (the groups run their epilogues side by side: an ordinary block barrier)
            auto chan = [&](float& n, float& mu, float& m2, float nb, float mub, float m2b) __attribute__((always_inline)) {
                const float d = mub - mu, nn = n + nb;
                mu += d * (nb / nn); m2 += m2b + d * d * (n * nb / nn); n = nn;
            };
            float mean_t[NR], m2_t[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const f32x2* e = red + (wr * 64 + row_of(r) + l15) * 4;
                float n = (float)WN, mu = e[0][0], m2 = e[0][1];
#pragma unroll
                for (int w = 1; w < 4; ++w) chan(n, mu, m2, (float)WN, e[w][0], e[w][1]);
                mean_t[r] = mu; m2_t[r] = m2;
            }
            // 4. publish + gather (round 6: no ticket, no counters -- the records carry their own validity).  Per row ONE 16-byte record {mean, tag, m2, tag} = two self-tagged
            //    8-byte granules, written through (sc1), tag = the tag of this tile's records of the op's previous launch + 1 (read at the top of this epilogue); the scratch
            //    is the op's own (mlblock.c wire_ln_fold), so a record with the right tag can only be this launch's.  ONE wave per (block, wave row) loads the partners' records with agent-scope loads until both tags of all of them match (bounded; every wave
            //    polling starved the arrivals themselves in round 3), hands them to the other waves through LDS; the others wait at the barrier.  The chain is now: record
            //    visible in L2 -> load returns it (was: store acknowledged -> ticket -> counter seen -> partials loaded), and there is nothing to reset.
            u32x4* gws = reinterpret_cast<u32x4*>(p.ln_ws) + ((long)rbk * nbn * BM);           // [tile column][128 rows] records
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)gws, 0, nbn * BM * 16, 0x00020000);
            if (wc == 0 && lg == 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const u32x4 rec = {__builtin_bit_cast(unsigned, mean_t[r]), tag, __builtin_bit_cast(unsigned, m2_t[r]), tag};
                    __builtin_amdgcn_raw_buffer_store_b128(rec, rs, (bnk * BM + wr * 64 + row_of(r) + l15) * 16, 0, 16);      // sc1: write-through
                }
            }
            // the fp32 output goes out while the other tiles' records arrive
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    *reinterpret_cast<f32x4*>(c32b + (long)row_of(r) * p.ldc32 + c * 16) = acc[r / RA][r % RA][c];
            f32x2* red2 = reinterpret_cast<f32x2*>(smem + 2 * STAGE + 4096);                   // [128 rows][8 tiles], beyond the ring and the wave columns' exchange
            if (wc == 0) {
                // lane (l15, lg): partner tiles lg and lg + 4, rows row_of(r) + l15 of this wave row.  (This tile's own record comes from registers.)
                const bool act0 = lg < nbn, act1 = lg + 4 < nbn;
                const bool own0 = lg == bnk, own1 = lg + 4 == bnk;
                u32x4 rec0[NR], rec1[NR];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    asm volatile("" ::: "memory");      // the record loads below are plain (readonly) buffer loads to the compiler: without this it hoists them out of the poll loop
                    if (act0 && !own0) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) rec0[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lg * BM + wr * 64 + row_of(r) + l15) * 16, 0, 16);      // sc1: past the vector L1
                    }
                    if (act1 && !own1) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) rec1[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((lg + 4) * BM + wr * 64 + row_of(r) + l15) * 16, 0, 16);
                    }
                    if (act0 && !own0) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) ok = ok && rec0[r][1] == tag && rec0[r][3] == tag;
                    }
                    if (act1 && !own1) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) ok = ok && rec1[r][1] == tag && rec1[r][3] == tag;
                    }
                    if (__all(ok) || ++spins >= (1u << 20)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (spins >= (1u << 20) && lane == 0) __hip_atomic_store(p.ln_cnt + 8191, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sticky give-up word (mlctx_handoff_check)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    f32x2* e = red2 + (wr * 64 + row_of(r) + l15) * 8;
                    if (act0) e[lg] = own0 ? f32x2{mean_t[r], m2_t[r]} : f32x2{__uint_as_float(rec0[r][0]), __uint_as_float(rec0[r][2])};
                    if (act1) e[lg + 4] = own1 ? f32x2{mean_t[r], m2_t[r]} : f32x2{__uint_as_float(rec1[r][0]), __uint_as_float(rec1[r][2])};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the ds_writes have LANDED before the barrier lets the readers through
            __builtin_amdgcn_s_barrier();
            // 5. all tiles' partials in tile order -> mean, 1 / sqrt(var + eps)
            float mean_r[NR], rstd_r[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const f32x2* e = red2 + (wr * 64 + row_of(r) + l15) * 8;
                f32x2 t = e[0];
                float n = (float)BN, mu = t[0], m2 = t[1];
                for (int b = 1; b < nbn; ++b) { t = e[b]; chan(n, mu, m2, (float)BN, t[0], t[1]); }
                mean_r[r] = mu; rstd_r[r] = 1.0f / sqrtf(m2 / n + p.ln_eps);
            }
            // 6. y = (v - mean) rstd gamma + beta -> fp16, 16-byte stores (pairs of column blocks exchange halves, as epi_fast)
            f32x4 gm[NCB], bt[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                gm[c] = *reinterpret_cast<const f32x4*>(p.ln_g + wcol0 + c * 16 + 4 * lg);
                bt[c] = *reinterpret_cast<const f32x4*>(p.ln_b + wcol0 + c * 16 + 4 * lg);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int qa = r / RA, i = r % RA;
                _Float16* rowp = p.ln_y + (long)(wrow0 + l15 + row_of(r)) * p.ldln + wcol0 + 8 * (lg >> 1) + 16 * (lg & 1);
                auto y4 = [&](int c) __attribute__((always_inline)) {
                    const f32x4 y = (acc[qa][i][c] - mean_r[r]) * rstd_r[r] * gm[c] + bt[c];
                    return f16x4{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
                };
#pragma unroll
                for (int c = 0; c + 1 < NCB; c += 2) {
                    const u32x2 a = __builtin_bit_cast(u32x2, y4(c)), b = __builtin_bit_cast(u32x2, y4(c + 1));
                    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    *reinterpret_cast<u32x4*>(rowp + c * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
                if constexpr (NCB & 1)
                    *reinterpret_cast<f16x4*>(p.ln_y + (long)(wrow0 + l15 + row_of(r)) * p.ldln + wcol0 + (NCB - 1) * 16 + 4 * lg) = y4(NCB - 1);
            }
        }
    };
    auto epi_xattn = [&](int wrow0, int wcol0) __attribute__((always_inline)) {
        if constexpr (BM == 128 && BN == 320 && !CONV && !SK) {
            (void)wrow0; (void)wcol0;
            constexpr int QP = 640, VP = 208;                        // LDS row pitches: q / K rows are 640 B with the ring's XOR swizzle of the 16-byte chunks inside every 128-byte segment
                                                                     // (chunk c of row r at slot c ^ ((r >> 1) & 7): b128 fragment reads of 16 rows touch every bank once); V^T rows 192 B of keys + 16
            constexpr int KROWS = 77;                                // K rows kept (Tk <= 77 real keys; rows beyond Tk are clamped duplicates, masked below)
            constexpr int QOFF = 57344;                              // q image behind K: the K image is 77 x 640 = 49 280 B, its 7 x 512 DMA slots cover 56 KB (the surplus lanes re-fetch the last chunk)
            constexpr int KCH = KROWS * 40, VCH = 320 * 13;          // 16-byte chunks of the K / V^T images
            constexpr int KIT = (KCH + 511) / 512, VIT = (VCH + 511) / 512;      // LDS-DMA instructions per thread: 7, 9
            static_assert(KIT * 512 * 16 <= QOFF && QOFF + 128 * QP <= 160 * 1024 && VIT * 512 * 16 <= 128 * QP, "LDS map");
            const int img = tcur.m0 / p.xa_Tq;
            const int Tk = p.xa_Tk;
            auto xstamp = [&](int slot) __attribute__((always_inline)) {       // diagnostics (tools/xattn_bench.py): a second block of 8 stamps per block behind the loop's 4096 x 8
                if (p.tbuf && tid == 0 && blockIdx.x < 4096) p.tbuf[32768 + (long)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
            };
            xstamp(0);
            // 1. ring drained (every wave's own tail DMAs, then everybody's), K on its way
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            xstamp(1);
            {
                const _Float16* kb = p.xa_k + (long)img * Tk * p.xa_ldk + tcur.n0;
#pragma unroll
                for (int it = 0; it < KIT; ++it) {
                    const int q = min(it * 512 + tid, KCH - 1);                        // (every lane of every instruction loads: the counted wait below counts instructions)
                    const int row = q / 40, sl = q % 40;                               // LDS slot sl of row `row` holds the chunk whose swizzled position it is
                    const int cc = (sl & ~7) | ((sl & 7) ^ ((row >> 1) & 7));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + (long)min(row, Tk - 1) * p.xa_ldk + cc * 8),      // (rows >= Tk: duplicates of the last key)
                                                     (__attribute__((address_space(3))) void*)(smem + (it * 512 + wave * 64) * 16), 16, 0, 0);
                }
            }
            // 2. q -> fp16 -> LDS [row][320 columns]
            {
                // row = wr WM + qa WM/2 + 16 i + l15 (swizzle term l15 >> 1), columns wc WN + 16 c + 4 lg .. + 3 = chunk 10 wc + 2 c + (lg >> 1), byte 8 (lg & 1) of it
                unsigned char* qw = smem + QOFF + (wr * WM + l15) * QP + (lg & 1) * 8;
                const int sw = l15 >> 1;
#pragma unroll
                for (int qa = 0; qa < 2; ++qa)
#pragma unroll
                    for (int i = 0; i < RA; ++i)
#pragma unroll
                        for (int c = 0; c < NCB; ++c) {
                            const f32x4 v = acc[qa][i][c] + cb[c];
                            const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                            const int ch = wc * (WN / 8) + 2 * c + (lg >> 1);
                            *reinterpret_cast<f16x4*>(qw + (qa * (WM / 2) + i * 16) * QP + (((ch & ~7) | ((ch & 7) ^ sw)) << 4)) = h;
                        }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the ds_writes have LANDED before the barrier lets the readers through (a raw s_barrier waits for nothing)
            __builtin_amdgcn_s_barrier();
            xstamp(2);
            // 3. this wave's q fragments: rows 16 wave + l15, head h, k-step s (natural k order: d = 32 s + 8 lg + e)
            f16x8 qf[5][2];
            {
                const unsigned char* qr = smem + QOFF + (wave * 16 + l15) * QP;
                const int sw = l15 >> 1;
#pragma unroll
                for (int h = 0; h < 5; ++h)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) qf[h][s2] = *reinterpret_cast<const f16x8*>(qr + h * 128 + (((4 * s2 + lg) ^ sw) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // fragments in registers: the q image may be overwritten
            __builtin_amdgcn_s_barrier();
            xstamp(3);
            {
                const _Float16* vb = p.xa_vt + ((long)img * p.N + tcur.n0) * 96;
#pragma unroll
                for (int it = 0; it < VIT; ++it) {
                    const int q = min(it * 512 + tid, VCH - 1);
                    const int row = q / 13, cc = min(q % 13, 11);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + (long)row * 96 + cc * 8),
                                                     (__attribute__((address_space(3))) void*)(smem + QOFF + (it * 512 + wave * 64) * 16), 16, 0, 0);
                }
            }
            wait_vmcnt<VIT>();                                        // K landed (this wave's part; the V^T instructions are younger)
            __builtin_amdgcn_s_barrier();
            xstamp(4);
            // 4. scores + softmax.  Every stage runs over ALL five heads before the next one starts: a head's chain (fragment reads -> MFMAs -> max -> exchange -> exp -> sum ->
            //    exchange -> pack) is ~1300 clocks of latency end to end, and with two waves per SIMD nothing else hides it -- as five interleaved independent chains the
            //    stages cost their issue time (in-kernel stamps, tools/xattn_bench.py: 6.7k -> clocks per block for this phase).  P stays in registers (3 k-steps of 32
            //    keys per head; keys 80..95 are zero).
            f16x8 pf[5][3];
            float linv[5];
            f32x4 sa[5][5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                const int krow = min(16 * kt + l15, KROWS - 1);              // (rows >= Tk of the image hold duplicates of the last key: masked below)
                const int ksw = (krow >> 1) & 7;
#pragma unroll
                for (int h = 0; h < 5; ++h) {
                    const unsigned char* kr = smem + krow * QP + h * 128;
                    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(kr + (((4 * s2 + lg) ^ ksw) << 4)), qf[h][s2], a4, 0, 0, 0);
                    sa[h][kt] = a4;
                }
            }
            // keys of tile kt in this lane: 16 kt + 4 lg + e
            {
                const int kfirst = 4 * lg;
#pragma unroll
                for (int kt = 0; kt < 5; ++kt) {
                    if (kt < 4 && Tk > 64) continue;                     // (the text context: only the last key tile is ragged; wave-uniform test)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool dead = 16 * kt + kfirst + e >= Tk;
#pragma unroll
                        for (int h = 0; h < 5; ++h) if (dead) sa[h][kt][e] = -3.0e38f;
                    }
                }
            }
            // v (op) the value of lane ^ 16 / lane ^ 32 without an LDS round trip: v_permlane16_swap / v_permlane32_swap exchange odd rows (the upper half) of their first
            // operand with even rows (the lower half) of the second; on two copies of v both results together hold the pair.  hipcc (ROCm 7.2) needs two opaque fences here:
            // handed the same value twice it may give both tied operands one register (the instruction then swaps a register with itself), and an arithmetic use of BOTH
            // results folds to the first one (`r[0] + r[1]` compiled to v + v: seen in the ISA, and as softmax sums off by a constant factor per row) -- the empty asm
            // statements make the copy and the two results distinct values for the optimiser and emit nothing but one v_mov.
            auto xor16 = [&](float v, auto op) __attribute__((always_inline)) {
                unsigned c;
                asm("v_mov_b32 %0, %1" : "=v"(c) : "v"(v));
                const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), c, false, false);
                unsigned r0 = r[0], r1 = r[1];
                asm volatile("" : "+v"(r0), "+v"(r1));
                return op(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
            };
            auto xor32 = [&](float v, auto op) __attribute__((always_inline)) {
                unsigned c;
                asm("v_mov_b32 %0, %1" : "=v"(c) : "v"(v));
                const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), c, false, false);
                unsigned r0 = r[0], r1 = r[1];
                asm volatile("" : "+v"(r0), "+v"(r1));
                return op(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
            };
            auto fmx = [](float a, float b) __attribute__((always_inline)) { return fmaxf(a, b); };
            auto fad = [](float a, float b) __attribute__((always_inline)) { return a + b; };
            float msc[5];
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                f32x4 m4 = sa[h][0];
#pragma unroll
                for (int kt = 1; kt < 5; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m4[e] = fmaxf(m4[e], sa[h][kt][e]);
                msc[h] = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            }
#pragma unroll
            for (int h = 0; h < 5; ++h) msc[h] = xor16(msc[h], fmx);
#pragma unroll
            for (int h = 0; h < 5; ++h) msc[h] = -xor32(msc[h], fmx) * p.xa_sc;
            float lsum[5];
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float pe = __builtin_amdgcn_exp2f(fmaf(sa[h][kt][e], p.xa_sc, msc[h])); sa[h][kt][e] = pe; s4[e] += pe; }
                lsum[h] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
#pragma unroll
            for (int h = 0; h < 5; ++h) lsum[h] = xor16(lsum[h], fad);
#pragma unroll
            for (int h = 0; h < 5; ++h) linv[h] = 1.0f / xor32(lsum[h], fad);
#pragma unroll
            for (int h = 0; h < 5; ++h)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    f16x8 f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[e] = (_Float16)sa[h][2 * ks][e];
                        f[4 + e] = (2 * ks + 1 < 5) ? (_Float16)sa[h][(2 * ks + 1 < 5) ? 2 * ks + 1 : 0][e] : (_Float16)0.f;
                    }
                    pf[h][ks] = f;
                }
            xstamp(5);
            wait_vmcnt<0>();                                          // V^T landed
            __builtin_amdgcn_s_barrier();
            xstamp(6);
            // 5. O^T = V^T . P, normalised, fp16; pairs of 16-column blocks exchange halves so that every lane stores 16 contiguous bytes (as epi_fast).  Per head: the 24
            //    8-byte fragment reads first, then 4 independent accumulation chains (one per 16 columns of d), k-step outermost
            _Float16* orow = p.xa_out + (long)(tcur.m0 + wave * 16 + l15) * p.xa_ldo + tcur.n0 + 8 * (lg >> 1) + 16 * (lg & 1);
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                union VF { f16x4 h4[2]; f16x8 f; };
                VF vf[4][3];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const unsigned char* vr = smem + QOFF + (h * 64 + dt * 16 + l15) * VP + lg * 8;
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        vf[dt][ks].h4[0] = *reinterpret_cast<const f16x4*>(vr + ks * 64);            // keys 32 ks + 4 lg ..
                        vf[dt][ks].h4[1] = *reinterpret_cast<const f16x4*>(vr + ks * 64 + 32);       // keys 32 ks + 16 + 4 lg ..
                    }
                }
                f32x4 oa[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) oa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dt][ks].f, pf[h][ks], oa[dt], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < 4; dt += 2) {
                    const f32x4 o0 = oa[dt] * linv[h], o1 = oa[dt + 1] * linv[h];
                    const f16x4 h0 = {(_Float16)o0[0], (_Float16)o0[1], (_Float16)o0[2], (_Float16)o0[3]};
                    const f16x4 h1 = {(_Float16)o1[0], (_Float16)o1[1], (_Float16)o1[2], (_Float16)o1[3]};
                    const u32x2 a = __builtin_bit_cast(u32x2, h0), b = __builtin_bit_cast(u32x2, h1);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    *reinterpret_cast<u32x4*>(orow + h * 64 + dt * 16) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
            }
            xstamp(7);
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int wrow0 = tcur.m0 + wr * WM, wcol0 = tcur.n0 + wc * WN;
        if (wrow0 >= p.M || wcol0 >= p.N) return;             // M % WM == 0, N % WN == 0: a wave's block is all in or all out
        const float* rbias = p.rowbias ? p.rowbias + (long)(tcur.m0 / p.rows_per_batch) * p.ldrb : nullptr;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int n = wcol0 + c * 16 + 4 * lg;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + n);
            if (rbias) v += *reinterpret_cast<const f32x4*>(rbias + n);
            cb[c] = v;
        }
        using std::integral_constant;
        using T = std::true_type;
        using F = std::false_type;
        if constexpr (EPI == PP_EPI_F16) { epi_fast(F{}, T{}, F{}, F{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_F32) { epi_fast(T{}, F{}, F{}, F{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_F32_RES) { epi_fast(T{}, F{}, T{}, F{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_F32_STATS) { epi_fast(T{}, F{}, F{}, T{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_F32_RES_STATS) { epi_fast(T{}, F{}, T{}, T{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_GEGLU16) { epi_fast_geglu(wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_F32_LN) { epi_ln(F{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_F32_RES_LN) { epi_ln(T{}, wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_XATTN) { epi_xattn(wrow0, wcol0); return; }
        if constexpr (EPI == PP_EPI_GENERIC) {
        if constexpr (RESBATCH) {
            if (p.resid) {
#pragma unroll
                for (int qa = 0; qa < 2; ++qa)
#pragma unroll
                    for (int i = 0; i < RA; ++i)
#pragma unroll
                        for (int c = 0; c < NCB; ++c)
                            rpre[(qa * RA + i) * NCB + c] = *reinterpret_cast<const f32x4*>(
                                p.resid + (long)(wrow0 + qa * (WM / 2) + i * 16 + l15) * p.ldr + wcol0 + c * 16 + 4 * lg);
            }
        }
        epi_rows(integral_constant<int, 0>{}, integral_constant<int, 0>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 0>{}, integral_constant<int, 1>{}, wrow0, wcol0);
        if constexpr (RA > 2) {
            epi_rows(integral_constant<int, 0>{}, integral_constant<int, 2>{}, wrow0, wcol0);
            epi_rows(integral_constant<int, 0>{}, integral_constant<int, 3>{}, wrow0, wcol0);
        }
        epi_rows(integral_constant<int, 1>{}, integral_constant<int, 0>{}, wrow0, wcol0);
        epi_rows(integral_constant<int, 1>{}, integral_constant<int, 1>{}, wrow0, wcol0);
        if constexpr (RA > 2) {
            epi_rows(integral_constant<int, 1>{}, integral_constant<int, 2>{}, wrow0, wcol0);
            epi_rows(integral_constant<int, 1>{}, integral_constant<int, 3>{}, wrow0, wcol0);
        }
        }
    };

    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (p.tbuf && tid == 0) p.tbuf[(long)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
    };
    stamp(0);
    // ---- prologue: U1..U4 of stream position 0 and U1 of position 1 in flight; U1(0), U2(0) retired and visible
    sa[0].par = sa[1].par = sb2.par = sb3.par = 0;
    enter_A(sa[0], 0); enter_A(sa[1], 1); enter_B2(sb2); enter_B3(sb3);
    kenter = 0;                                    // every later tile is entered at its first K tile
    tcur = tnext;
    tnext = tile_at(ntile > 1 ? 1 : 0);            // past the end: any valid tile (staged, never read)
    if constexpr (NPH == 4 && SCH == 1) {          // the steady-state issue order: U1, U2 (0) | U4 (0) | U3 (0) | U1, U2 (1)
        issue_A(sa[0], 0); issue_B2(); issue_A(sa[1], 1); issue_B3(); issue_A(sa[0], 0); issue_B2();
        wait_vmcnt<NU1 + NU2 + NU3 + NU1>();
    } else {
        issue_A(sa[0], 0); issue_B2(); issue_B3(); issue_A(sa[1], 1);
        if constexpr (NPH == 4) { issue_A(sa[0], 0); wait_vmcnt<W4>(); }      // U1 runs two K tiles ahead
        else wait_vmcnt<NU1>();                                               // two phases: U1..U3(0) landed, U4(0) retired by slot A
    }
    __builtin_amdgcn_s_barrier();
    stamp(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one barrier behind group 0 from here on

    // timing-only BUILDS of the loop (make EXTRA=-DMLSD_PP_DBG=n, tools/gemm_trace.py; the results are wrong): 4 = no LDS-DMA issue in the loop,
    // 8 = no fragment reads, 16 = no MFMAs, 32 = no s_setprio.  Compile-time: the same switches as run-time tests cost the production loop 3-59
    // spilled registers.
#ifndef MLSD_PP_DBG
#define MLSD_PP_DBG 0
#endif
    constexpr int dbg = MLSD_PP_DBG;
#define MLSD_PP_PHASE(QA, QB, LOAD_A, LOAD_B, ISSUE, WAITN)                                                  \
    {                                                                                                        \
        if (!(dbg & 8)) {                                                                                    \
            if (LOAD_B) read_B(stage, QB);                                                                   \
            if (LOAD_A) read_A(stage, QA);                                                                   \
        }                                                                                                    \
        if (!(dbg & 4)) { ISSUE; }                                                                           \
        if (WAITN >= 0) wait_vmcnt<(WAITN >= 0 ? WAITN : 0)>();                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (!(dbg & 32)) __builtin_amdgcn_s_setprio(1);                                                      \
        if (!(dbg & 16)) {                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                   \
                _Pragma("unroll") for (int c = (QB ? CB0 : 0); c < (QB ? NCB : CB0); ++c)                    \
                    acc[QA][i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[c][ks], af[i][ks], acc[QA][i][c], 0, 0, 0); \
        }                                                                                                    \
        if (!(dbg & 32)) __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

    // ---- stream-K hand-off (SK): accumulators <-> this block's slab, [wave][register group][lane] f32x4: 1 KiB per wave-instruction
    auto sk_publish = [&]() __attribute__((always_inline)) {
        if constexpr (SK) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sk_ws + (long)skw * (BM * BN)), 0, BM * BN * 4, 0x00020000);
            int idx = wave * (2 * RA * NCB) * 64 + lane;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < RA; ++i)
#pragma unroll
                    for (int c = 0; c < NCB; ++c, idx += 64)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][i][c]), rs, idx * 16, 0, 16);   // aux 16 = sc1: write-through
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains (and with it the staging loads in flight: one bubble per contribution)
            __builtin_amdgcn_s_barrier();
            if (tid == 0) __hip_atomic_store(p.sk_flag + skw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto sk_gather = [&](int end_k) __attribute__((always_inline)) {   // owner of an unfinished tile: add the slabs of the blocks holding K tiles end_k .. nkt-1
        if constexpr (SK) {
            // Round 4: ALL contributors' flags first, then the slabs GB at a time per accumulator row group.  The first form walked the slabs one by one (flag, barrier,
            // 2 RA rounds of NCB loads each waiting for the previous): 4 exposed memory round trips per slab, ~4 us each -- with 5..14 contributors per tile (the small-M,
            // long-K convolutions of SD1.5: 2048x640x5760, 512x1280x11520) the gather was longer than the K loop.  Here a row group costs ceil(nc / GB) round trips for all slabs.
            // The order of the additions per accumulator is unchanged (owner's part, then contributors in block order): results are bit-identical to the first form.
            if constexpr (BM == 256) {
            // (256 x 256 tile: 128 accumulator registers leave no room for a second slab in flight -- the batched form below spills 84..136 bytes per lane there -- so it keeps
            // the slab-by-slab walk; its launches have 2..3 contributors per tile)
            int c = skw + 1;
            for (int rem = nkt - end_k; rem > 0; rem -= p.sk_L, ++c) {
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.sk_flag + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(4);
                    if (spins >= (1u << 22)) __hip_atomic_store(p.sk_flag + 4095, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(p.sk_flag + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one consumer per flag: clear it for the next launch
                }
                __builtin_amdgcn_s_barrier();
                const f32x4* src = reinterpret_cast<const f32x4*>(p.sk_ws + (long)c * (BM * BN)) + wave * (2 * RA * NCB) * 64 + lane;
                // NCB loads in flight at a time: left free, the compiler issued all 2 RA NCB loads (128 registers) beside the 128
                // accumulators and spilled through the whole kernel (5x slower main loop)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < RA; ++i) {
                        f32x4 t[NCB];
#pragma unroll
                        for (int cc = 0; cc < NCB; ++cc) t[cc] = src[((a * RA + i) * NCB + cc) * 64];
#pragma unroll
                        for (int cc = 0; cc < NCB; ++cc) acc[a][i][cc] += t[cc];
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            } else {
            const int nc = (nkt - end_k + p.sk_L - 1) / p.sk_L;        // contributors: blocks skw + 1 .. skw + nc
            if (tid < nc && tid < 64) {
                unsigned spins = 0;                                // bounded: a lost contribution must not hang the device; the give-up is REPORTED through the
                // sticky word sk_flag[4095], which the host reads with the results (mlctx_handoff_check): a contributor block that is not resident (CUs taken
                // by another process) would otherwise turn into a silently wrong tile
                while (__hip_atomic_load(p.sk_flag + skw + 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(4);
                if (spins >= (1u << 22)) __hip_atomic_store(p.sk_flag + 4095, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.sk_flag + skw + 1 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one consumer per flag: clear it for the next launch
            }
            for (int c = 64; c < nc; ++c) {                        // (more than 64 contributors per tile: never chosen by sk_share's >= 4 K tiles per block on real shapes, but correct)
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(p.sk_flag + skw + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(4);
                    if (spins >= (1u << 22)) __hip_atomic_store(p.sk_flag + 4095, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.sk_flag + skw + 1 + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // program order only: the slab loads below must not be hoisted above the flag poll / barrier by the compiler
            // the slabs are read with agent-scope (sc1) loads: they were written through by blocks of other XCDs (no AGENT acquire fence: invalidating the caches of 8 waves costs more)
            constexpr int GB = 4;                                                  // slabs in flight per row group: GB * NCB float4 beside the accumulators
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sk_ws + (long)(skw + 1) * (BM * BN)), 0, 0x7fffffff, 0x00020000);
            const int lofs = (wave * (2 * RA * NCB) * 64 + lane) * 16;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    for (int c0 = 0; c0 < nc; c0 += GB) {
                        f32x4 t[GB][NCB];
#pragma unroll
                        for (int b = 0; b < GB; ++b) {
                            const int cs = min(c0 + b, nc - 1);           // (past the last slab: re-read it, never added)
#pragma unroll
                            for (int cc = 0; cc < NCB; ++cc)
                                t[b][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lofs + ((a * RA + i) * NCB + cc) * 64 * 16, cs * (BM * BN * 4), 16));
                        }
#pragma unroll
                        for (int b = 0; b < GB; ++b) {
                            if (c0 + b < nc) {
#pragma unroll
                                for (int cc = 0; cc < NCB; ++cc) acc[a][i][cc] += t[b][cc];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

#define MLSD_PP_PHASE2(QA, READS, ISSUE, WAITN)                                                              \
    {                                                                                                        \
        READS;                                                                                               \
        ISSUE;                                                                                               \
        wait_vmcnt<WAITN>();                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                   \
                _Pragma("unroll") for (int c = 0; c < NCB; ++c)                                              \
                    acc[QA][i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[c][ks], af[i][ks], acc[QA][i][c], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }

    int kt = kt0, ti = 0;
    for (int s = 0; s < S; ++s) {
        const unsigned char* stage = smem + (s & 1) * STAGE;
        if constexpr (NPH == 2) {
            MLSD_PP_PHASE2(0, (read_B(stage, 0), read_B(stage, 1), read_A(stage, 0)), (issue_A(sa[0], 0), issue_B2(), issue_B3()), NU1 + NU2 + NU3)
            MLSD_PP_PHASE2(1, read_A(stage, 1), issue_A(sa[1], 1), NU1)
        } else if constexpr (SCH == 1) {
        MLSD_PP_PHASE(0, 0, true, true, issue_A(sa[1], 1), NU1 + NU2 + NU1)
        MLSD_PP_PHASE(0, 1, false, true, issue_B3(), -1)
        MLSD_PP_PHASE(1, 1, true, false, (void)0, -1)
        MLSD_PP_PHASE(1, 0, false, false, (issue_A(sa[0], 0), issue_B2()), NU1 + NU2 + NU3 + NU1)
        } else {
        MLSD_PP_PHASE(0, 0, true, true, issue_B2(), W1)
        MLSD_PP_PHASE(0, 1, false, true, issue_B3(), W2)
        MLSD_PP_PHASE(1, 1, true, false, issue_A(sa[1], 1), -1)
        MLSD_PP_PHASE(1, 0, false, false, issue_A(sa[0], 0), W4)
        }
        const bool tile_done = ++kt == nkt;
        if (tile_done || (SK && s == S - 1)) {
            // seam: the next output tile's first units are in flight / landed; nothing is drained.  The groups
            // rejoin (group 0's extra barrier pairs with group 1's last one) so that all 8 waves run their
            // epilogues side by side, then group 1 drops one barrier behind again (equal barrier counts).
            if (wr == 0) __builtin_amdgcn_s_barrier();
            if (ti == 0) stamp(2);
            if (ti == ntile - 1) stamp(4);
            if constexpr (SK) {
                const bool contrib = ti == 0 && kt0 > 0;               // this segment started inside the tile: a contribution
                if (!contrib && !tile_done) sk_gather(kt);             // this block owns the tile but its stream ends before the tile does
                if (contrib) sk_publish(); else epilogue();
            } else {
                if (!(p.dbg & 1)) epilogue();
            }
            if (ti == 0) stamp(3);
            if (ti == ntile - 1) stamp(5);
            zero_acc();
            kt = 0; ++ti;
            tcur = tnext;                                  // every sequence has entered tile ti by now (K >= 3 K tiles)
            tnext = tile_at(ti + 1 < ntile ? ti + 1 : 0);
            if (s + 1 < S && wr == 1) __builtin_amdgcn_s_barrier();
        }
    }
#undef MLSD_PP_PHASE
#undef MLSD_PP_PHASE2
    wait_vmcnt<0>();                               // the tail stages (clamped rows) still target this block's LDS
    stamp(6);
    if (p.tbuf && tid == 0) p.tbuf[(long)blockIdx.x * 8 + 7] = ntile;
}
