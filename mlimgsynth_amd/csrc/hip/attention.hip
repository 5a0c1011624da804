// Fused (flash-style) multi-head attention for gfx950 (MI355X), fp16 operands, fp32 accumulate.
//
// Replaces ggml_nn_attention (reference src/ggml_extend.c:200-222), which materialises the
// [Tk,Tq,heads] fp32 score tensor in memory (671 MB per SDXL layer), and the head split/merge
// permutes around it (src/mlblock_nn.c:204-227).  Here the scores never leave the CU:
//
//   block = 4 wavefronts = 128 query rows of one (batch, head); each wave owns 32 query rows.
//   per 64-key tile (K and V staged global -> registers -> LDS, double-buffered, one barrier per tile, shared by the 4 waves):
//     S^T[key][q] = K . Q^T          v_mfma_f32_32x32x16_f16, A = K rows from LDS (ds_read_b128),
//                                    B = Q fragments kept in registers for the whole kernel.
//                                    Swapped product: the key index lands in the 16 accumulator
//                                    registers and the query on the lane, so the softmax row
//                                    reduction is in-lane plus ONE cross-half exchange.
//     online softmax                 running max m / sum l per query (lane), exp2 with the
//                                    1/sqrt(d)*log2(e) scale folded in.
//     O^T[d][q] += V^T . P           the S^T accumulator registers are converted to fp16 and used
//                                    DIRECTLY as the B operand (cdna_hip_programming.md §3
//                                    "accumulator tile as the next MFMA's operand": permuted k
//                                    order, verified by tests/test_hw_probe.py); the matching
//                                    V^T fragments come from the row-major V tile in LDS through
//                                    ds_read_b64_tr_b16 (hardware transpose read, T10).
//   epilogue: O / l, heads merged, fp16 [T][n_head*d_head].
//
// d_head in {40, 64, 80, 160}: K columns are zero-padded to a multiple of 16 and V columns to a
// multiple of 32 in LDS.  Tk needs no alignment (77-token cross attention); causal mask optional.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "mlsd_kernels.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float max3f(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// two fp32 FMAs per VALU issue slot (v_pk_fma_f32): the softmax argument s*sc - m*sc of two scores at once
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp2_pair(float a, float b, float sc, float msc)
{
    const f32x2 t = __builtin_elementwise_fma(f32x2{a, b}, f32x2{sc, sc}, f32x2{msc, msc});
    return f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};   // raw v_exp_f32: arguments <= 6, underflow to 0 is the wanted result
}
// The same with two v_fma_f32 (attn64x2s_kernel).  Packed fp32 vector instructions (v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32; v_dot2_f32_f16 too) do NOT run in the shadow
// of an MFMA on gfx950: with a 32x32x16 MFMA in flight three of them cost more than the MFMA's 32 clocks ON TOP of it, while v_fma_f32, v_add_f32, v_exp_f32,
// v_cvt_pk_f16_f32, v_max3_f32 and v_pk_fma_f16 disappear behind it (tools/coissue_probe.py, profiles/r6_coissue_probe.txt).  The tile-loop kernels keep the packed forms:
// their vector phases mostly run with the matrix pipe idle, and both forms time the same there (profiles/r6_attn_packed_vs_scalar_ab.txt).  Same bits either way.
__device__ __forceinline__ f32x2 exp2_pair_s(float a, float b, float sc, float msc)
{
    float t0, t1;                                                            // (asm: the vectoriser would pair them again)
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(a), "v"(sc), "v"(msc));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(b), "v"(sc), "v"(msc));
    return f32x2{__builtin_amdgcn_exp2f(t0), __builtin_amdgcn_exp2f(t1)};
}

struct AttnP {
    const _Float16 *q, *k, *v;
    _Float16* o;
    long ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    int n_head, Tq, Tk, causal;
    int nq, G;  // query blocks per (batch, head) group; number of groups
    unsigned long long* tbuf;   // diagnostics (attn64pp_kernel): cycle stamps of block 0, [wave][tile][5], or null
    int wide_o; // output rows are 16-byte aligned: lane pairs exchange halves and store 16 bytes each (T21)
    float sc;  // 1/sqrt(d_head) * log2(e)
};

template <int DH, bool VSUM>
// waves_per_eu caps the occupancy the register allocator aims for: left free it chose 4 waves/SIMD (128 VGPRs) and
// spilled the K/V staging registers to scratch inside the loop
__global__ __launch_bounds__(256, (DH <= 80 ? 2 : 1)) __attribute__((amdgpu_waves_per_eu(1, (DH <= 80 ? 3 : 1)))) void attn_kernel(const AttnP p)
{
    constexpr int DQK = (DH + 15) / 16 * 16;       // QK^T reduction length (zero padded)
    constexpr int NKS = DQK / 16;
    constexpr int NDV = (DH + 31) / 32;            // 32-wide output tiles
    constexpr int KSTR = DQK * 2 + 16;             // bytes; stride/4 mod 64 = 4 mod 8 -> conflict-free b128 row reads
    constexpr int VSTR = ((NDV & 1) ? NDV : NDV + 1) * 64;  // bytes; odd multiple of 64 -> conflict-free tr reads
    constexpr int CH = DH / 8;                     // 16-byte chunks per K/V row
    constexpr int IT = (64 * CH + 255) / 256;
    static_assert(DH % 8 == 0, "d_head must be a multiple of 8");

    // two K/V tile buffers: tile t+1 is written (registers -> LDS) into the buffer tile t-1 vacated, so ONE barrier per
    // tile orders both the reads of tile t and the writes of tile t+1
    __shared__ __attribute__((aligned(16))) unsigned char Ks2[2][64 * KSTR];
    __shared__ __attribute__((aligned(16))) unsigned char Vs2[2][64 * VSTR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD-aware block -> (group, query block) map: consecutive workgroup ids go round-robin to the 8 XCDs, so id % 8 is the
    // XCD.  All query blocks of one (batch, head) group run on ONE XCD (group g -> XCD g % 8): that XCD's L2 fetches the
    // group's K/V once instead of each of the 8 L2s fetching it (the 3.4x fabric traffic measured in round 1).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gk = slot / p.nq, qb = slot - gk * p.nq;
    const int grp = gk * 8 + xcd;
    if (grp >= p.G) return;                          // padding blocks of a group count that is not a multiple of 8 (whole block leaves)
    const int head = grp % p.n_head, b = grp / p.n_head;
    const int q0 = qb * 128, qw = q0 + wave * 32;
    const int qrow = qw + lr;

    const _Float16* Qg = p.q + (long)b * p.bsq + (long)head * DH;
    const _Float16* Kg = p.k + (long)b * p.bsk + (long)head * DH;
    const _Float16* Vg = p.v + (long)b * p.bsv + (long)head * DH;

    // zero the LDS once: padding columns (d_head not a multiple of 16 / 32) stay zero for the whole kernel; at d_head 32, 64,
    // 160 every column the MFMAs read is data and the fill (a fixed cost per block that dominates 77-key cross attention) is skipped
    if constexpr (DQK != DH || NDV * 32 != DH) {
        for (int i = tid * 16; i < 2 * 64 * KSTR; i += 256 * 16) *reinterpret_cast<uint4*>(&Ks2[0][0] + i) = make_uint4(0, 0, 0, 0);
        for (int i = tid * 16; i < 2 * 64 * VSTR; i += 256 * 16) *reinterpret_cast<uint4*>(&Vs2[0][0] + i) = make_uint4(0, 0, 0, 0);
    }

    // Q fragments (B operand of S^T = K.Q^T): lane (lr = query, lh) holds Q[q][16*ks + 8*lh + j]
    f16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = 16 * ks + 8 * lh;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qrow < p.Tq && c < DH) v = *reinterpret_cast<const uint4*>(Qg + (long)qrow * p.ldq + c);
        qf[ks] = __builtin_bit_cast(f16x8, v);
    }

    f32x16 oacc[NDV];
#pragma unroll
    for (int d = 0; d < NDV; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -1.0e30f;
    // row sums l[q] = sum_k P[k][q] run on the matrix pipe: ones[32 x 16] . P puts the sum in every row of the result (64
    // v_add per tile and the final cross-half exchange disappear; the sum is over the fp16 P the P.V product uses)
    f32x16 lacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) lacc[e] = 0.f;
    f32x2 vsum = {0.f, 0.f};                        // VSUM: per-lane partial row sums on the VALU (v_pk_add_f32), see attn64x2_kernel
    const f16x8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};

    int nt = (p.Tk + 63) / 64;
    if (p.causal) { const int lim = (min(q0 + 128, p.Tq) + 63) / 64; nt = min(nt, lim); }

    uint4 rk[IT], rv[IT];
    // K/V rows past Tk are CLAMPED to the last key instead of zero-filled: their scores are masked to -1e30 (P = 0), so
    // finite duplicate data is as good as zeros and the loop carries no predication, zero fills or exec-mask branches.
    // Full tiles are fetched through running pointers (one 64-bit add per load); only a ragged last tile recomputes
    // clamped addresses.  (Threads past the tile's last chunk re-load the last chunk; loads go through temporaries:
    // assigned directly, the staging arrays were demoted to scratch/LDS by the compiler.)
    const _Float16* kp[IT];
    const _Float16* vp[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = (IT * 256 > 64 * CH) ? min(tid + it * 256, 64 * CH - 1) : tid + it * 256;
        kp[it] = Kg + (long)(idx / CH) * p.ldk + (idx % CH) * 8;
        vp[it] = Vg + (long)(idx / CH) * p.ldv + (idx % CH) * 8;
    }
    auto load_kv = [&](int t) {
        const int kv0 = t * 64;
        if (kv0 + 64 <= p.Tk) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const uint4 a = *reinterpret_cast<const uint4*>(kp[it] + (long)kv0 * p.ldk);
                const uint4 c = *reinterpret_cast<const uint4*>(vp[it] + (long)kv0 * p.ldv);
                rk[it] = a; rv[it] = c;
            }
        } else {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = (IT * 256 > 64 * CH) ? min(tid + it * 256, 64 * CH - 1) : tid + it * 256;
                const int row = idx / CH, ch = idx % CH;
                const long r = min(kv0 + row, p.Tk - 1);
                const uint4 a = *reinterpret_cast<const uint4*>(Kg + r * p.ldk + ch * 8);
                const uint4 c = *reinterpret_cast<const uint4*>(Vg + r * p.ldv + ch * 8);
                rk[it] = a; rv[it] = c;
            }
        }
    };
    auto store_kv = [&](int buf) {
        unsigned char* Ks = Ks2[buf];
        unsigned char* Vs = Vs2[buf];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / CH, ch = idx % CH;
            if (idx < 64 * CH) {
                *reinterpret_cast<uint4*>(Ks + row * KSTR + ch * 16) = rk[it];
                *reinterpret_cast<uint4*>(Vs + row * VSTR + ch * 16) = rv[it];
            }
        }
    };

    load_kv(0);
    if constexpr (DQK != DH || NDV * 32 != DH) __syncthreads();   // LDS zero-fill complete
    store_kv(0);
    __syncthreads();

    // transposed-read lane address pattern (probe-verified): group g = lane>>4, idx = lane&15
    const int tg = lane >> 4, ti = lane & 15;
    const int tr_row = 4 * (tg >> 1) + (ti >> 2);          // + key block base
    const int tr_col = 16 * (tg & 1) + 4 * (ti & 3);       // + 32*di

    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * 64;
        const unsigned char* Ks = Ks2[t & 1];
        const unsigned char* Vs = Vs2[t & 1];
        if (t + 1 < nt) load_kv(t + 1);

        // ---- S^T = K . Q^T  (two 32-key sub-tiles; sub1: the second holds at least one key -- when not, e.g. keys
        //      64..76 of the 77-token cross attention live in sub-tile 0 of tile 1, its P.V products are skipped)
        const bool sub1 = kv0 + 32 < p.Tk;                                   // wave-uniform
        // The two accumulators are separate SSA values whose MFMA chains start from an inline zero: as an array
        // initialised to zero ahead of a conditional chain the compiler materialised 32 zeros and copied 32 result
        // registers per tile (a quarter of the loop's VALU instructions).
        f32x16 sacc[2];
        {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 r0 = zero, r1 = zero;
            // both sub-tiles, also when the second holds no key (rows >= Tk are clamped duplicates; masked below): no
            // branch, no fill; the two accumulation chains alternate so that no MFMA waits for its predecessor's result
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const f16x8 k0 = *reinterpret_cast<const f16x8*>(Ks + lr * KSTR + (16 * ks + 8 * lh) * 2);
                const f16x8 k1 = *reinterpret_cast<const f16x8*>(Ks + (32 + lr) * KSTR + (16 * ks + 8 * lh) * 2);
                r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[ks], r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[ks], r1, 0, 0, 0);
            }
            sacc[0] = r0; sacc[1] = r1;
        }
        // ---- online softmax over the key axis (registers + one cross-half exchange).
        // Raw scores stay unscaled: p = exp2(s*sc - m*sc) is ONE fma + v_exp per element.  Masking only in
        // tiles that need it (tail of Tk, causal diagonal): the test is wave-uniform.
        const bool need_mask = (kv0 + 64 > p.Tk) || (p.causal && kv0 + 63 > qw);
        if (need_mask) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (key >= p.Tk || (p.causal && key > qrow)) sacc[kt][e] = -1.0e30f;
                }
        }
        // max through v_max3_f32 (asm): plain fmaxf on MFMA outputs costs a canonicalising v_max per operand
        float mx = max3f(sacc[0][0], sacc[0][1], sacc[1][0]);
        mx = max3f(mx, sacc[1][1], sacc[0][2]);
#pragma unroll
        for (int e = 3; e < 16; e += 2) mx = max3f(mx, sacc[0][e], (e + 1 < 16) ? sacc[0][e + 1] : sacc[0][e]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) mx = max3f(mx, sacc[1][e], sacc[1][e + 1]);
        mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);
        // deferred rescale (T13): keep the old reference maximum while the new one exceeds it by less than
        // 2^6 in the exp2 domain (P <= 64 fits fp16 with full relative precision); the decision is taken for
        // the whole wave, BEFORE this tile's P is formed, so O, l and P always share one reference.
        const bool grow = (mx - m_run) * p.sc > 6.0f;
        if (__any(grow)) {
            const float m_new = max3f(m_run, mx, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
            m_run = m_new;
            if constexpr (VSUM) vsum *= alpha;
            else {
#pragma unroll
                for (int e = 0; e < 16; ++e) lacc[e] *= alpha;
            }
#pragma unroll
            for (int d = 0; d < NDV; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }
        const float msc = -m_run * p.sc;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 r = exp2_pair(sacc[kt][e], sacc[kt][e + 1], p.sc, msc);
                sacc[kt][e] = r.x; sacc[kt][e + 1] = r.y;
                if constexpr (VSUM) vsum += r;
            }

        // ---- O^T += V^T . P   (P = S^T accumulators as B operand, permuted k order)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1 && !sub1) continue;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (_Float16)sacc[kt][8 * s + j];
                const int kb = 32 * kt + 16 * s;
                if constexpr (!VSUM) lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, lacc, 0, 0, 0);   // row sums on the matrix pipe (ones . P)
#pragma unroll
                for (int d = 0; d < NDV; ++d) {
                    const unsigned char* a0 = Vs + (kb + tr_row) * VSTR + (32 * d + tr_col) * 2;
                    union { h16x4 h[2]; f16x8 f; } vf;
                    vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)a0);
                    vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(a0 + 8 * VSTR));
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.f, pf, oacc[d], 0, 0, 0);
                }
            }
        }
        // keep the registers -> LDS copy HERE (a full tile of MFMA/softmax work after the loads were issued): left free,
        // the scheduler hoists it up to the loads and waits for them on the spot
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) store_kv((t + 1) & 1);   // the buffer tile t-1 used: every wave passed the previous barrier after reading it
        __syncthreads();                         // tile t+1 visible; every wave is done reading tile t
    }

    // ---- epilogue: O = O^T / l, heads merged.  oacc[d][e]: d-index = 32*d + (e&3) + 8*(e>>2) + 4*lh, query = lane&31
    float l_fin = lacc[0];
    if constexpr (VSUM) { const float l = vsum.x + vsum.y; l_fin = l + __shfl_xor(l, 32, 64); }
    const float inv = 1.0f / l_fin;
    if (qrow < p.Tq) {
        _Float16* og = p.o + (long)b * p.bso + (long)qrow * p.ldo + (long)head * DH;
        auto piece = [&](int d, int eg) __attribute__((always_inline)) {
            const f16x4 h = {(_Float16)(oacc[d][4 * eg + 0] * inv), (_Float16)(oacc[d][4 * eg + 1] * inv),
                             (_Float16)(oacc[d][4 * eg + 2] * inv), (_Float16)(oacc[d][4 * eg + 3] * inv)};
            return __builtin_bit_cast(u32x2, h);
        };
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int eg = 0; eg < 4; eg += 2) {
                // a row's 8 columns 32d + 8eg .. +7 sit as two 8-byte pieces in lanes q and q + 32: pairs of them swap halves
                // (v_permlane32_swap) so that every lane stores 16 contiguous bytes -- the tail is store-issue bound
                // (cdna_hip_programming.md T21); needs both column groups inside d_head and 16-byte aligned rows
                if (p.wide_o && 32 * d + 8 * eg + 16 <= DH) {
                    u32x2 a = piece(d, eg), c = piece(d, eg + 1);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                    *reinterpret_cast<u32x4*>(og + 32 * d + 8 * eg + 8 * lh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                } else {
#pragma unroll
                    for (int e2 = eg; e2 < eg + 2; ++e2) {
                        const int dbase = 32 * d + 8 * e2 + 4 * lh;
                        if (dbase < DH) *reinterpret_cast<u32x2*>(og + dbase) = piece(d, e2);
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// d_head = 64, no causal mask, Tq a multiple of 256 (every attention of the SDXL UNet): the throughput kernel.
//   block = 4 waves = 256 query rows; a wave owns TWO 32-row query blocks (64 rows): every K fragment read from LDS feeds
//   two QK^T MFMAs and every transposed V fragment two P.V MFMAs (half the LDS reads per FLOP of the kernel above), and
//   the two blocks are independent dependency chains the scheduler interleaves (MFMA of one under the softmax VALU of the
//   other).  K/V tiles (64 keys) go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write),
//   double buffered, one barrier per tile.  Unpadded 128-byte rows; the bank-conflict XOR is applied to the SOURCE chunk:
//     K image: slot = chunk ^ ((row >> 1) & 7)        conflict-free ds_read_b128 of 32 rows x one chunk (as the GEMM tiles)
//     V image: slot = chunk ^ (((row >> 1) & 1) << 2) conflict-free ds_read_b64_tr_b16 of 4 rows x 64 B per 32-lane half
//   Keys past Tk (77-token cross attention) are clamped duplicates of the last key and masked.
//   VSUM: row sums l on the VALU (v_pk_add_f32 over the fp32 P, one cross-half exchange at the end) instead of the ones . P
//   MFMAs (8 of the 40 MFMAs per tile): trades 256 matrix-pipe cycles for 128 VALU cycles per tile.
template <bool VSUM>
__global__ __launch_bounds__(256, 2) void attn64x2_kernel(const AttnP p)
{
    constexpr int DH = 64, RB = 128;                 // bytes per K/V row
    __shared__ __attribute__((aligned(1024))) unsigned char Ks2[2][64 * RB];
    __shared__ __attribute__((aligned(1024))) unsigned char Vs2[2][64 * RB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gk = slot / p.nq, qb = slot - gk * p.nq;
    const int grp = gk * 8 + xcd;
    if (grp >= p.G) return;
    const int head = grp % p.n_head, b = grp / p.n_head;
    const int qw = qb * 256 + wave * 64;

    const _Float16* Qg = p.q + (long)b * p.bsq + (long)head * DH;
    const _Float16* Kg = p.k + (long)b * p.bsk + (long)head * DH;
    const _Float16* Vg = p.v + (long)b * p.bsv + (long)head * DH;

    // Q fragments of both query blocks (B operand of S^T = K.Q^T): lane (lr = query, lh) holds Q[q][16*ks + 8*lh + j]
    f16x8 qf[2][4];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[sb][ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Qg + (long)(qw + 32 * sb + lr) * p.ldq + 16 * ks + 8 * lh));

    const int nt = (p.Tk + 63) / 64;
    // LDS-DMA staging: wave w fills rows 16w .. 16w+15 of the tile with two 1-KiB pieces (8 rows each) per operand
    const int srow = lane >> 3, sslot = lane & 7;
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + srow;
            const long key = min(t * 64 + row, p.Tk - 1);
            const int ck = sslot ^ ((row >> 1) & 7), cv = sslot ^ (((row >> 1) & 1) << 2);
            unsigned char* dk = Ks2[buf] + (wave * 16 + i * 8) * RB;      // wave-uniform piece base, lane-linear image
            unsigned char* dv = Vs2[buf] + (wave * 16 + i * 8) * RB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kg + key * p.ldk + ck * 8),
                                             (__attribute__((address_space(3))) void*)dk, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vg + key * p.ldv + cv * 8),
                                             (__attribute__((address_space(3))) void*)dv, 16, 0, 0);
        }
    };

    f32x16 o0[2], o1[2];                              // O^T accumulators of query block 0 / 1: [d-block of 32]
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[d][e] = 0.f; o1[d][e] = 0.f; }
    float m0 = -1.0e30f, m1 = -1.0e30f;
    // row sums l[q] = sum_k P[k][q] accumulated on the matrix pipe: ones[32 x 16] . P -> every row of the result holds the sums
    f32x16 ls0, ls1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ls0[e] = 0.f; ls1[e] = 0.f; }
    f32x2 vs0 = {0.f, 0.f}, vs1 = {0.f, 0.f};         // VSUM: per-lane partial row sums (two interleaved chains)
    const f16x8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing.  K rows lr / 32+lr have the same swizzle term ((32 + lr) >> 1) & 7 == (lr >> 1) & 7.
    const int kswz = (lr >> 1) & 7;
    const int tg = lane >> 4, ti = lane & 15;
    const int tr_row = 4 * (tg >> 1) + (ti >> 2);                    // + key block base (multiple of 8): swizzle term from tr_row alone
    const int vswz = ((tr_row >> 1) & 1) << 2;
    const int tr_c = 2 * (tg & 1) + ((ti & 3) >> 1), tr_b = 8 * (ti & 1);   // 16-byte chunk (+4*d) and byte inside it

    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * 64;
        const unsigned char* Ks = Ks2[t & 1];
        const unsigned char* Vs = Vs2[t & 1];
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);   // the buffer tile t-1 used: every wave passed the barrier that ended iteration t-1

        // ---- S^T = K . Q^T for both query blocks: 8 K fragment reads feed 16 MFMAs
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 s00 = zero, s01 = zero, s10 = zero, s11 = zero;        // s<block><key sub-tile>
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((2 * ks + lh) ^ kswz) << 4;
            const f16x8 k0 = *reinterpret_cast<const f16x8*>(Ks + lr * RB + off);
            const f16x8 k1 = *reinterpret_cast<const f16x8*>(Ks + (32 + lr) * RB + off);
            s00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[0][ks], s00, 0, 0, 0);
            s10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[1][ks], s10, 0, 0, 0);
            s01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[0][ks], s01, 0, 0, 0);
            s11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[1][ks], s11, 0, 0, 0);
        }
        if (kv0 + 64 > p.Tk) {                        // ragged last tile (wave-uniform): mask the clamped duplicates
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (key >= p.Tk) { s00[e] = -1.0e30f; s10[e] = -1.0e30f; }
                if (key + 32 >= p.Tk) { s01[e] = -1.0e30f; s11[e] = -1.0e30f; }
            }
        }
        // ---- online softmax of one query block (registers + one cross-half exchange; deferred rescale, see attn_kernel)
        // max as v_max3_f32 through asm: a plain fmaxf on MFMA outputs makes hipcc insert a canonicalising v_max per operand
        // (100 v_max for 64 scores, MI355X_MICROARCH.md); row sums run on the matrix pipe (P . ones, below) instead of 64 v_add
        auto softmax = [&](f32x16& sa, f32x16& sb_, float& m_run, f32x16* oacc, f32x16& lacc, f32x2& vsum) __attribute__((always_inline)) {
            float mx = max3f(sa[0], sa[1], sb_[0]);
            mx = max3f(mx, sb_[1], sa[2]);
#pragma unroll
            for (int e = 3; e < 16; e += 2) mx = max3f(mx, sa[e], (e + 1 < 16) ? sa[e + 1] : sa[e]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = max3f(mx, sb_[e], sb_[e + 1]);
            mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);
            const bool grow = (mx - m_run) * p.sc > 6.0f;
            if (__any(grow)) {
                const float m_new = max3f(m_run, mx, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
                m_run = m_new;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
                if constexpr (VSUM) vsum *= alpha;
                else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) lacc[e] *= alpha;
                }
            }
            const float msc = -m_run * p.sc;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 ra = exp2_pair(sa[e], sa[e + 1], p.sc, msc), rb = exp2_pair(sb_[e], sb_[e + 1], p.sc, msc);
                sa[e] = ra.x; sa[e + 1] = ra.y; sb_[e] = rb.x; sb_[e + 1] = rb.y;
                if constexpr (VSUM) vsum += ra + rb;
            }
        };
        softmax(s00, s01, m0, o0, ls0, vs0);
        softmax(s10, s11, m1, o1, ls1, vs1);

        // ---- O^T += V^T . P for both query blocks: 16 transposed V reads feed 16 MFMAs
        const bool sub1 = kv0 + 32 < p.Tk;            // the second 32-key sub-tile holds at least one key (wave-uniform)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1 && !sub1) continue;
            const f32x16& pa = kt ? s01 : s00;
            const f32x16& pb = kt ? s11 : s10;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                f16x8 pf0, pf1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { pf0[j] = (_Float16)pa[8 * sx + j]; pf1[j] = (_Float16)pb[8 * sx + j]; }
                const int kb = 32 * kt + 16 * sx;
                if constexpr (!VSUM) {
                    ls0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf0, ls0, 0, 0, 0);
                    ls1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf1, ls1, 0, 0, 0);
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* a0 = Vs + (kb + tr_row) * RB + (((4 * d + tr_c) ^ vswz) << 4) + tr_b;
                    union { h16x4 h[2]; f16x8 f; } vf;
                    vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)a0);
                    vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(a0 + 8 * RB));
                    o0[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.f, pf0, o0[d], 0, 0, 0);
                    o1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.f, pf1, o1[d], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t+1 have landed
        __syncthreads();                                    // all pieces visible; every wave is done reading tile t
    }

    // ---- epilogue: O = O^T / l, heads merged.  o[d][e]: d-index = 32*d + (e&3) + 8*(e>>2) + 4*lh, query = lane&31
    auto store = [&](const f32x16* oacc, float l_run, int sb) __attribute__((always_inline)) {
        const float inv = 1.0f / l_run;                 // the MFMA row sum already covers all 64 keys of every tile (both lane halves)
        _Float16* og = p.o + (long)b * p.bso + (long)(qw + 32 * sb + lr) * p.ldo + (long)head * DH;
        auto piece = [&](int d, int eg) __attribute__((always_inline)) {
            const f16x4 h = {(_Float16)(oacc[d][4 * eg + 0] * inv), (_Float16)(oacc[d][4 * eg + 1] * inv),
                             (_Float16)(oacc[d][4 * eg + 2] * inv), (_Float16)(oacc[d][4 * eg + 3] * inv)};
            return __builtin_bit_cast(u32x2, h);
        };
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int eg = 0; eg < 4; eg += 2) {
                if (p.wide_o) {             // 16 bytes per lane (see attn_kernel's epilogue)
                    u32x2 a = piece(d, eg), c = piece(d, eg + 1);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                    *reinterpret_cast<u32x4*>(og + 32 * d + 8 * eg + 8 * lh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                } else {
                    *reinterpret_cast<u32x2*>(og + 32 * d + 8 * eg + 4 * lh) = piece(d, eg);
                    *reinterpret_cast<u32x2*>(og + 32 * d + 8 * (eg + 1) + 4 * lh) = piece(d, eg + 1);
                }
            }
    };
    if constexpr (VSUM) {
        const float l0 = vs0.x + vs0.y, l1 = vs1.x + vs1.y;         // a lane half holds the keys 4*lh + {0..3} + 8j of every 32
        store(o0, l0 + __shfl_xor(l0, 32, 64), 0);
        store(o1, l1 + __shfl_xor(l1, 32, 64), 1);
    } else {
        store(o0, ls0[0], 0);
        store(o1, ls1[0], 1);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// attn64x2s (round 6): the 64-rows-per-wave kernel above, SOFTWARE-PIPELINED inside the wave.  d_head 64, Tq % 256 == 0, Tk % 64 == 0, Tk >= 128 (the self attentions of SDXL).
//
// Why.  At d = 64 a 64-key tile of one 32-row query block is 16 MFMAs (512 matrix clocks) against ~550 issue clocks of softmax (32 v_exp_f32 at 8, 16 packed FMAs, 16 packed
// adds, 16 converts, 18 max3), and in the tile-loop kernels the two streams ADD: the compiler's schedule has the 16 QK^T MFMAs of both query blocks first, then block 0's whole
// softmax with the matrix pipe idle, then block 1's softmax partly under block 0's P.V (~1240 clocks per 32-row wave-tile whatever the occupancy: profiles/NOTES.md).  What
// does overlap is a wave's OWN vector instructions in the shadow of its own MFMAs (an MFMA holds the issue port for a fraction of its 32 clocks: MI355X_MICROARCH.md).
//
// Here the two query blocks (A, B) of a wave run HALF A STEP APART on 32-key sub-tiles (u = 0, 1 of tile t), and every quarter pairs one block's softmax of 16 scores per lane
// with 8 MFMAs of the OTHER block, which do not depend on it:
//     Q1(t):  softmax A(t,0)   |   QK^T B(t,0)    (4)  +  P.V B(t-1,1)  (4)        fragments X(t)  = K(t) u 0, V(t-1) u 1   (second use)
//     Q2(t):  softmax B(t,0)   |   QK^T A(t,1)    (4)  +  P.V A(t,0)    (4)        fragments Y(t)  = K(t) u 1, V(t) u 0     (first use)
//     Q3(t):  softmax A(t,1)   |   QK^T B(t,1)    (4)  +  P.V B(t,0)    (4)        fragments Y(t)                            (second use)
//     Q4(t):  softmax B(t,1)   |   QK^T A(t+1,0)  (4)  +  P.V A(t,1)    (4)        fragments X(t+1) = K(t+1) u 0, V(t) u 1  (first use)
// written as 8 slices of {1 MFMA, at most one fragment read, 1/8 of the softmax} with a scheduling barrier between slices, so that the emitted stream IS the interleave
// (maximum + rescale decision in the first slice, then 8 exp chunks of 2 scores).  Every K / V fragment is read from LDS ONCE and serves both query blocks in two consecutive
// quarters (32 registers), as in the tile-loop kernel: a first form that re-read them per block issued twice the LDS reads.  The reads of a
// fragment set are issued one per slice in the quarter BEFORE its first use, into the registers the previous set leaves one slice earlier (the last one in the first slice of the
// first use): seven slices of flight.
// LDS reads and their waits are inline asm.  hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 it emits from the builtin while an LDS-DMA is
// in flight (the intrinsic carries no alias information, so it "may alias" the DMA's target) -- in the tile-loop kernels that is one early wait per tile, here it exposed a
// whole global round trip twice per tile (the first build of this kernel ran at HALF the tile loop's speed for that reason alone).  The counted lgkmcnt waits take the fragment
// as an in/out operand so that no use can be scheduled above them; counts are the LDS instructions issued after the fragment (any the compiler adds only make the wait longer).
// K / V tiles go global -> LDS by LDS-DMA into rings of three buffers: tile t + 2 is issued at the top of iteration t into the buffer tile t - 1 left, drained (`vmcnt(0)`, a
// whole tile of compute later) and published by the ONE barrier at the end of the iteration; tile t + 1's first fragments are read in Q3(t).
// Same fragment layouts and per-element arithmetic as the tile-loop kernels (row sums on the VALU), but the running maximum is revisited every 32 keys instead of every 64.
#ifndef SP_PREMAX
#define SP_PREMAX 0         // 1 = the sub-tile's maximum already in slices 5-7 of the quarter whose slices 0-3 produce the scores (shortens the next quarter's first slice).
                            // MEASURED NOT REPEATABLE: the v_max3_f32 there are inline asm (max3f), the compiler places no wait states in front of an asm's reads, and two
                            // slices behind the last QK^T MFMA they sometimes see the accumulator before its final write (first-use quarters, whose MFMAs start late behind
                            // their fragment waits): 1-ulp differences between runs in block A only (tools/attn_sp_debug.py).  In the first slice of the NEXT quarter the
                            // producing MFMAs are seven matrix instructions back in an in-order pipe.  No speed difference either way.
#endif
#ifndef SP_QSCALE
#define SP_QSCALE 1         // 1 = Q fragments pre-multiplied by log2(e) / sqrt(d) (rounded to fp16 once more) and -m as the C operand of the first QK^T MFMA: the accumulators ARE the
                            // exp2 arguments, no v_fma_f32 per score (16 of ~75 vector instructions of a quarter); 0 = the tile-loop kernels' form (fp32 scale after the MFMA)
#endif
#ifndef SP_SHFL
#define SP_SHFL 0           // diagnostic: 1 = the lane halves exchange their maximum through ds_bpermute
#endif
#ifndef SP_ABL
#define SP_ABL 0            // diagnostic builds (tools/attn_sp_ablate.sh): 1 = no exp, 2 = no MFMA, 4 = no fragment reads, 8 = no maximum / rescale, 16 = no fp32 -> fp16 conversion of P: WRONG results, timing only
#endif
template <int I_> using sp_ic = std::integral_constant<int, I_>;
template <int B_, int E_, class F_> __device__ __forceinline__ void sp_for(F_&& f) { if constexpr (B_ < E_) { f(sp_ic<B_>{}); sp_for<B_ + 1, E_>(f); } }
template <int OFF> __device__ __forceinline__ void sp_read_b128(f16x8& r, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void sp_read_tr64(u32x2& r, unsigned a) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); }
template <int N> __device__ __forceinline__ void sp_wait(f16x8& r) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N)); }
template <int N> __device__ __forceinline__ void sp_wait2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// DHV = the head dimension: 64, or 40 (SD1.5's 4096-token level) run AS 64 -- the Q fragments are zero beyond column 40, so whatever the K rows hold there (the next
// head's columns, or a clamped duplicate at the end of the row) multiplies zero; the V columns beyond 40 produce output rows that are never stored.  5 of 8 matrix
// instructions of a QK^T / P.V pair do useful work then: still 1.3 - 1.5 x the general kernel at d = 40 (which pads to 48 / 64 as well).
template <int DHV>
__global__ __launch_bounds__(256, 2) void attn64x2s_kernel(const AttnP p)
{
    constexpr int DH = DHV, RB = 128;
    __shared__ __attribute__((aligned(1024))) unsigned char Ks3[3][64 * RB];
    __shared__ __attribute__((aligned(1024))) unsigned char Vs3[3][64 * RB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gk = slot / p.nq, qb = slot - gk * p.nq;
    const int grp = gk * 8 + xcd;
    if (grp >= p.G) return;
    const int head = grp % p.n_head, b = grp / p.n_head;
    const int qw = qb * 256 + wave * 64;

    const _Float16* Qg = p.q + (long)b * p.bsq + (long)head * DH;
    const _Float16* Kg = p.k + (long)b * p.bsk + (long)head * DH;
    const _Float16* Vg = p.v + (long)b * p.bsv + (long)head * DH;

    f16x8 qf[2][4];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 qv = make_uint4(0, 0, 0, 0);
            if (DHV == 64 || 16 * ks + 8 * lh + 8 <= DHV) qv = *reinterpret_cast<const uint4*>(Qg + (long)(qw + 32 * sb + lr) * p.ldq + 16 * ks + 8 * lh);
            qf[sb][ks] = __builtin_bit_cast(f16x8, qv);
            if constexpr (SP_QSCALE != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[sb][ks][j] = (_Float16)((float)qf[sb][ks][j] * p.sc);
            }
        }

    const int nt = p.Tk / 64;
    const int srow = lane >> 3, sslot = lane & 7;
    // (DHV < 64: a row of a head is DHV * 2 / 16 chunks of data; the chunks behind it are the following heads' columns -- never read past the last head's)
    const int cmax = DHV == 64 ? 7 : min(7, (p.n_head - head) * (DHV * 2 / 16) - 1);
    // LDS-DMA of this wave's 16 rows of a tile: two 1-KiB pieces per operand (chunk swizzles applied to the SOURCE, as in the kernel above)
    auto stage = [&](int t, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 16 + i * 8 + srow;
            const int ck = min(sslot ^ ((row >> 1) & 7), cmax), cv = min(sslot ^ (((row >> 1) & 1) << 2), cmax);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kg + (long)(t * 64 + row) * p.ldk + ck * 8),
                                             (__attribute__((address_space(3))) void*)(Ks3[buf] + (wave * 16 + i * 8) * RB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vg + (long)(t * 64 + row) * p.ldv + cv * 8),
                                             (__attribute__((address_space(3))) void*)(Vs3[buf] + (wave * 16 + i * 8) * RB), 16, 0, 0);
        }
    };

    f32x16 oA[2], oB[2];                              // O^T accumulators of query block A / B: [d-block of 32]
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { oA[d][e] = 0.f; oB[d][e] = 0.f; }
    float mA = -1.0e30f, mB = -1.0e30f, mxA = 0.f, mxB = 0.f;
    // SP_QSCALE: -m' of a block's queries as the 16-register C operand of its QK^T (m' = the reference maximum in the exp2 domain, 0 before the first sub-tile); the growth
    // threshold of the deferred rescale: -3e38 before the block's first sub-tile (its maximum becomes the reference whatever it is), 6 afterwards
    f32x16 negA, negB;
#pragma unroll
    for (int e = 0; e < 16; ++e) { negA[e] = 0.f; negB[e] = 0.f; }
    float thrA = -3.0e38f, thrB = -3.0e38f;
    f16x8 pfA[2], pfB[2];                             // P of one 32-key sub-tile as B operands: keys 16 sx ..
    const f16x8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
    pfA[0] = pfA[1] = pfB[0] = pfB[1] = hzero;

    // fragment addressing (byte offsets inside a 64-row tile image; LDS addresses are 32-bit)
    const int kswz = (lr >> 1) & 7;
    const int tg = lane >> 4, ti = lane & 15;
    const int tr_row = 4 * (tg >> 1) + (ti >> 2);
    const int vswz = ((tr_row >> 1) & 1) << 2;
    const int tr_c = 2 * (tg & 1) + ((ti & 3) >> 1), tr_b = 8 * (ti & 1);
    unsigned koff[4], voff[2];                        // K: row lr, k-step ks;  V: row tr_row, d-block d   (+ 4096 u, + 2048 sx, + 1024 for the second half: immediates)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(lr * RB + (((2 * ks + lh) ^ kswz) << 4));
#pragma unroll
    for (int d = 0; d < 2; ++d) voff[d] = (unsigned)(tr_row * RB + (((4 * d + tr_c) ^ vswz) << 4) + tr_b);
    const unsigned ks_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&Ks3[0][0];
    const unsigned vs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&Vs3[0][0];

    // the fragment set in use: kf[ks] = K rows of a sub-tile (A operand of QK^T, k-step ks); vf[j] = V^T of 16 keys x 32 d (A operand of P.V; j = 2 sx + d), read as two halves
    f16x8 kf[4], vf[4];
    u32x2 vlo[4], vhi[4];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // read fragment F (0..3: K k-step F; 4..7: V piece F - 4) of sub-tile U: K from the tile image at kb, V from the one at vb
    auto issue = [&](auto F_c, auto UK_c, auto UV_c, unsigned kb, unsigned vb) __attribute__((always_inline)) {
        constexpr int F = decltype(F_c)::value, UK = decltype(UK_c)::value, UV = decltype(UV_c)::value;
        if constexpr (F < 4) sp_read_b128<4096 * UK>(kf[F], kb + koff[F]);
        else {
            constexpr int j = F - 4;
            const unsigned a = vb + voff[j & 1];
            sp_read_tr64<4096 * UV + 2048 * (j >> 1)>(vlo[j], a);
            sp_read_tr64<4096 * UV + 2048 * (j >> 1) + 1024>(vhi[j], a);
        }
    };
    // LDS instructions issued after fragment F of a set when its first use waits for it: fragments F + 1 .. 6 were issued before the quarter (K: 1 instruction, V: 2), fragment 7
    // in the quarter's first slice, after its MFMA
    auto younger = [](int F) constexpr { int n = 0; for (int g = F + 1; g <= 6; ++g) n += g < 4 ? 1 : 2; if (F >= 1 && F <= 6) n += 2; return n; };

    // Row sums l[q] on the matrix pipe (the vector unit is the busier one here, and the packed adds that would halve their cost do not run beside an MFMA: exp2_pair_s above):
    // v_mfma_f32_16x16x32_f16 with P as the B operand -- its lane (n = lane & 15, k group g = lane >> 4) is the 32x32x16 P fragment's lane (query n + 16 (g & 1), key half
    // g >> 1) -- and a 0 / 1 selector as A: output row m takes the k groups with g & 1 == (m >> 2) & 1, so that lane l (rows 4 (l >> 4) .., column l & 15) ends with the sum
    // of ITS OWN query l & 31 in every element (the rescale factor of a lane then applies to the sum it holds).  16 matrix clocks and 4 accumulator registers per block.
    const _Float16 selv = (((lane >> 4) & 1) == (((lane & 15) >> 2) & 1)) ? (_Float16)1.f : (_Float16)0.f;
    const f16x8 sel = {selv, selv, selv, selv, selv, selv, selv, selv};
    f32x4 lsA = {0.f, 0.f, 0.f, 0.f}, lsB = {0.f, 0.f, 0.f, 0.f};
    // the other lane half's value without an LDS round trip (v_permlane32_swap on two copies; the fences: see gemm_pp.hpp xor32)
    auto half_max = [&](float v) __attribute__((always_inline)) {
        if constexpr (SP_SHFL != 0) return max3f(v, __shfl_xor(v, 32, 64), v);
        unsigned c, a = __builtin_bit_cast(unsigned, v);
        asm("v_mov_b32 %0, %1" : "=v"(c) : "v"(a));
        const auto r = __builtin_amdgcn_permlane32_swap(a, c, false, false);
        unsigned r0 = r[0], r1 = r[1];
        asm volatile("" : "+v"(r0), "+v"(r1));
        return max3f(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1), __builtin_bit_cast(float, r1));
    };

    // One quarter.  Softmax of `sc` (16 scores per lane of one block's 32-key sub-tile; mxS = their maximum over the sub-tile, found in the quarter that produced them)
    // -> pfS, beside the other block's MFMAs on the fragment set in registers: QK^T of its next sub-tile (qfM -> sM, maximum -> mxM), P.V of its pending P (pfM, oM) and that
    // P's row sums (lsM).  FIRST: the set's first use (wait for every fragment; fragment 7 is read in slice 0); otherwise the next set is read behind this one: fragment
    // i - 1 in slice i (K sub-tile UK of the image at kb, V sub-tile UV of the image at vb).
    auto quarter = [&](auto FIRST_c, auto UK_c, auto UV_c, unsigned kb, unsigned vb,
                       f32x16& sc, float mxS, float& m_run, f32x16& negS, float& thrS, f32x16* oS, f32x4& lsS, f16x8* pfS,
                       const f16x8* qfM, f32x16& sM, float& mxM, const f32x16& negM, f32x16* oM, f32x4& lsM, const f16x8* pfM) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(FIRST_c)::value;
        float msc = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        sp_for<0, 8>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (FIRST && !(SP_ABL & 4)) {
                if constexpr (i < 4) sp_wait<younger(i)>(kf[i]);
                else {
                    sp_wait2<younger(i)>(vlo[i - 4], vhi[i - 4]);
                    vf[i - 4] = __builtin_bit_cast(f16x8, u32x4{vlo[i - 4][0], vlo[i - 4][1], vhi[i - 4][0], vhi[i - 4][1]});
                }
            }
            if constexpr (SP_ABL & 2) { }
            else if constexpr (i == 0) sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qfM[0], SP_QSCALE ? negM : zero, 0, 0, 0);
            else if constexpr (i < 4) sM = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[i], qfM[i], sM, 0, 0, 0);
            else {
                oM[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[i - 4], pfM[(i - 4) >> 1], oM[i & 1], 0, 0, 0);
                if constexpr (i & 1) lsM = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, pfM[(i - 4) >> 1], lsM, 0, 0, 0);
            }
            if constexpr (SP_ABL & 4) { }
            else if constexpr (FIRST) { if constexpr (i == 0) issue(sp_ic<7>{}, UK_c, UV_c, kb, vb); }
            else { if constexpr (i >= 1) issue(sp_ic<i - 1>{}, UK_c, UV_c, kb, vb); }
            if constexpr (i == 0 && !(SP_ABL & 8)) {
                if constexpr (!SP_PREMAX) {
                    // one asm statement: the compiler puts a wait state behind every asm it cannot see into (7 s_nop per quarter with one v_max3_f32 per statement)
                    float a1, a2, a3, a4;
                    asm("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %10, %11, %12\n\tv_max3_f32 %3, %13, %14, %15\n\t"
                        "v_max3_f32 %0, %0, %16, %17\n\tv_max3_f32 %1, %1, %18, %19\n\tv_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %3"
                        : "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4)
                        : "v"(sc[0]), "v"(sc[1]), "v"(sc[2]), "v"(sc[3]), "v"(sc[4]), "v"(sc[5]), "v"(sc[6]), "v"(sc[7]), "v"(sc[8]), "v"(sc[9]), "v"(sc[10]), "v"(sc[11]),
                          "v"(sc[12]), "v"(sc[13]), "v"(sc[14]), "v"(sc[15]));
                    mxS = half_max(a1);
                }
                if constexpr (SP_QSCALE != 0) {
                    // the scores ARE exp2 arguments relative to the block's reference maximum: mxS > 6 (or the block's first sub-tile) moves the reference
                    const bool grow = mxS > thrS;
                    if (__any(grow)) {
                        const float delta = fmaxf(mxS, thrS - 6.0f);                       // first sub-tile: mxS itself, afterwards max(mxS, 0)
                        const float alpha = __builtin_amdgcn_exp2f(fminf(-delta, 64.0f));   // (first sub-tile: O = l = 0, the factor only has to stay finite)
#pragma unroll
                        for (int d = 0; d < 2; ++d)
#pragma unroll
                            for (int e = 0; e < 16; ++e) oS[d][e] *= alpha;
                        lsS *= alpha;
#pragma unroll
                        for (int e = 0; e < 16; ++e) { sc[e] -= delta; negS[e] -= delta; }
                    }
                    thrS = 6.0f;
                } else {
                const bool grow = (mxS - m_run) * p.sc > 6.0f;       // deferred rescale, as attn_kernel: decided for the whole wave before this sub-tile's P is formed
                if (__any(grow)) {
                    const float m_new = max3f(m_run, mxS, mxS);
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
                    m_run = m_new;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oS[d][e] *= alpha;
                    lsS *= alpha;
                }
                msc = -m_run * p.sc;
                }
            }
            {   // exp chunk i: scores 2 i, 2 i + 1; after chunks 3 and 7 the 8 finished scores become one B fragment (the accumulator registers in their permuted k order)
                const f32x2 r = (SP_ABL & 1) ? f32x2{sc[2 * i] * p.sc + msc, sc[2 * i + 1] * p.sc + msc}
                              : SP_QSCALE ? f32x2{__builtin_amdgcn_exp2f(sc[2 * i]), __builtin_amdgcn_exp2f(sc[2 * i + 1])} : exp2_pair_s(sc[2 * i], sc[2 * i + 1], p.sc, msc);
                float r0 = r.x, r1 = r.y;
                asm volatile("" : "+v"(r0), "+v"(r1));          // the chunk stays in its slice: without a side effect the optimiser sinks a quarter's exps below the next quarter's first slice
                sc[2 * i] = r0; sc[2 * i + 1] = r1;
                if constexpr ((i & 3) == 3) {
                    f16x8 f;
                    if constexpr (SP_ABL & 16) f = __builtin_bit_cast(f16x8, u32x4{__builtin_bit_cast(unsigned, sc[8 * (i >> 2)]), __builtin_bit_cast(unsigned, sc[8 * (i >> 2) + 1]), __builtin_bit_cast(unsigned, sc[8 * (i >> 2) + 2]), __builtin_bit_cast(unsigned, sc[8 * (i >> 2) + 3])});
                    else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (_Float16)sc[8 * (i >> 2) + j];
                    }
                    asm volatile("" : "+v"(f));
                    pfS[i >> 2] = f;
                }
            }
            if constexpr (!(SP_ABL & 8) && SP_PREMAX) {         // the maximum of the scores slices 0-3 produced, for the quarter that will exponentiate them
                if constexpr (i == 5) { m1 = max3f(sM[0], sM[1], sM[2]); m2 = max3f(sM[3], sM[4], sM[5]); m3 = max3f(sM[6], sM[7], sM[8]); m4 = max3f(sM[9], sM[10], sM[11]); asm volatile("" : "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4)); }
                if constexpr (i == 6) { m1 = max3f(m1, sM[12], sM[13]); m2 = max3f(m2, sM[14], sM[15]); m1 = max3f(m1, m2, m3); m1 = max3f(m1, m4, m4); asm volatile("" : "+v"(m1)); }
                if constexpr (i == 7) { mxM = half_max(m1); asm volatile("" : "+v"(mxM)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using U0 = sp_ic<0>; using U1 = sp_ic<1>;

    // ---- prologue: tiles 0 and 1 staged and landed; X(0) = K(0) u 0 (V(-1) does not exist: zero fragments against a zero P); QK^T A(0,0)
    stage(0, 0); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    sp_for<0, 4>([&](auto f_c) __attribute__((always_inline)) { issue(f_c, U0{}, U0{}, ks_base, vs_base); });
    sp_for<0, 4>([&](auto f_c) __attribute__((always_inline)) { constexpr int f = decltype(f_c)::value; sp_wait<3 - f>(kf[f]); vf[f] = hzero; });
    f32x16 sA = zero, sB = zero;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sA = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[0][ks], sA, 0, 0, 0);
    {
        float m1 = max3f(sA[0], sA[1], sA[2]), m2 = max3f(sA[3], sA[4], sA[5]), m3 = max3f(sA[6], sA[7], sA[8]), m4 = max3f(sA[9], sA[10], sA[11]);
        m1 = max3f(m1, sA[12], sA[13]); m2 = max3f(m2, sA[14], sA[15]);
        m1 = max3f(m1, m2, m3);
        mxA = half_max(max3f(m1, m4, m4));
    }

    int cur = 0;                                              // ring slot of tile t
    for (int t = 0; t < nt; ++t) {
        const int nx1 = cur == 2 ? 0 : cur + 1, prv = cur == 0 ? 2 : cur - 1;      // slots of tiles t + 1 and t - 1 (the one tile t + 2 goes to)
        if (t + 2 < nt) stage(t + 2, prv);
        const unsigned kc = ks_base + cur * (64 * RB), vc = vs_base + cur * (64 * RB), kn = ks_base + nx1 * (64 * RB);
        // Q1: softmax A(t,0) | B on X(t);  reads Y(t) = K(t) u 1, V(t) u 0
        quarter(F_{}, U1{}, U0{}, kc, vc, sA, mxA, mA, negA, thrA, oA, lsA, pfA, qf[1], sB, mxB, negB, oB, lsB, pfB);
        // Q2: softmax B(t,0) | A on Y(t) (first use)
        quarter(T_{}, U1{}, U0{}, kc, vc, sB, mxB, mB, negB, thrB, oB, lsB, pfB, qf[0], sA, mxA, negA, oA, lsA, pfA);
        // Q3: softmax A(t,1) | B on Y(t);  reads X(t+1) = K(t+1) u 0, V(t) u 1
        quarter(F_{}, U0{}, U1{}, kn, vc, sA, mxA, mA, negA, thrA, oA, lsA, pfA, qf[1], sB, mxB, negB, oB, lsB, pfB);
        // Q4: softmax B(t,1) | A on X(t+1) (first use; past the last tile K(t+1) is whatever the slot holds: scores nobody reads)
        quarter(T_{}, U0{}, U1{}, kn, vc, sB, mxB, mB, negB, thrB, oB, lsB, pfB, qf[0], sA, mxA, negA, oA, lsA, pfA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t + 2, issued a tile of compute ago
        __syncthreads();                                      // tile t + 2 visible; every wave is done reading tile t - 1's successor slot ... and tile t
        cur = nx1;
    }
    // ---- P.V of block B, last sub-tile (V(nt-1) u 1: the V half of the set in registers)
#pragma unroll
    for (int j = 0; j < 4; ++j) oB[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[j], pfB[j >> 1], oB[j & 1], 0, 0, 0);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) lsB = __builtin_amdgcn_mfma_f32_16x16x32_f16(sel, pfB[sx], lsB, 0, 0, 0);

    auto store = [&](const f32x16* oacc, float l_run, int sb) __attribute__((always_inline)) {
        const float inv = 1.0f / l_run;
        _Float16* og = p.o + (long)b * p.bso + (long)(qw + 32 * sb + lr) * p.ldo + (long)head * DH;
        auto piece = [&](int d, int eg) __attribute__((always_inline)) {
            const f16x4 h = {(_Float16)(oacc[d][4 * eg + 0] * inv), (_Float16)(oacc[d][4 * eg + 1] * inv),
                             (_Float16)(oacc[d][4 * eg + 2] * inv), (_Float16)(oacc[d][4 * eg + 3] * inv)};
            return __builtin_bit_cast(u32x2, h);
        };
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int eg = 0; eg < 4; eg += 2) {
                if (32 * d + 8 * eg >= DHV) continue;                  // (DHV = 40: columns 40 .. 63 do not exist)
                if (p.wide_o) {
                    u32x2 a = piece(d, eg), c = piece(d, eg + 1);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                    if (32 * d + 8 * eg + 8 * lh + 8 <= DHV) *reinterpret_cast<u32x4*>(og + 32 * d + 8 * eg + 8 * lh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                } else {
                    *reinterpret_cast<u32x2*>(og + 32 * d + 8 * eg + 4 * lh) = piece(d, eg);
                    if (32 * d + 8 * (eg + 1) + 8 <= DHV) *reinterpret_cast<u32x2*>(og + 32 * d + 8 * (eg + 1) + 4 * lh) = piece(d, eg + 1);
                }
            }
    };
    store(oA, lsA[0], 0);
    store(oB, lsB[0], 1);
}


// ---------------------------------------------------------------------------------------------------------------------
// Tk <= 96, no causal mask (the 77-token cross attention of every UNet layer): ONE pass.
//   The general kernels walk 64-key tiles with an online softmax: for 77 keys that is two tiles (128 key slots, 28 MFMAs and
//   ~340 vector instructions per 32 query rows), two barriers and a rescale test per block, and the launch is bound by that
//   per-block chain (tools/attn_bench.py: 20.8 us for 42 MB).  Here the whole key set (padded to 96 = three 32-key sub-tiles,
//   rows past Tk are clamped duplicates whose scores are masked) is staged ONCE per workgroup; a wave then walks QB 32-row
//   query blocks: S^T = K.Q^T for the sub-tiles that hold keys, one max / exp2 / sum over the lane's <= 48 scores, O^T += V^T.P
//   for the 16-key steps that hold keys (77 keys: 12 + 10 MFMAs, 40 exponentials per lane), no loop over keys, no barrier and
//   no rescale branch in the block loop; the next block's Q is fetched as soon as the current QK^T has consumed the registers.
//   Same fragment layouts as attn_kernel (swapped QK^T, S accumulators reused as the P.V B operand, V^T by ds_read_b64_tr_b16).
template <int DH, int WPS>       // WPS: waves per SIMD the register allocation aims for
__global__ __launch_bounds__(256, WPS) void attn_tk96_kernel(const AttnP p, const int QB)
{
    constexpr int DQK = (DH + 15) / 16 * 16, NKS = DQK / 16, NDV = (DH + 31) / 32;
    constexpr int KSTR = DQK * 2 + 16;             // bytes; conflict-free b128 row reads (see attn_kernel)
    constexpr int VSTR = ((NDV & 1) ? NDV : NDV + 1) * 64;
    constexpr int CH = DH / 8, NK = 96;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[NK * KSTR];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[NK * VSTR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;         // all blocks of one (batch, head) group on one XCD (see attn_kernel)
    const int gk = slot / p.nq, qb = slot - gk * p.nq;
    const int grp = gk * 8 + xcd;
    if (grp >= p.G) return;
    const int head = grp % p.n_head, b = grp / p.n_head;
    const _Float16* Qg = p.q + (long)b * p.bsq + (long)head * DH;
    const _Float16* Kg = p.k + (long)b * p.bsk + (long)head * DH;
    const _Float16* Vg = p.v + (long)b * p.bsv + (long)head * DH;
    const int Tk = p.Tk;

    auto load_q = [&](f16x8 (&qf)[NKS], int qrow) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int c = 16 * ks + 8 * lh;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qrow < p.Tq && c < DH) v = *reinterpret_cast<const uint4*>(Qg + (long)qrow * p.ldq + c);
            qf[ks] = __builtin_bit_cast(f16x8, v);
        }
    };
    const int q_first = qb * QB * 128 + wave * 32 + lr;
    f16x8 qf[NKS];
    load_q(qf, q_first);                                             // in flight while the keys are staged

    // K / V (rows >= Tk: duplicates of the last key) -> LDS; padding columns (d_head not a multiple of 16 / 32) zeroed
    if constexpr (DQK != DH || NDV * 32 != DH) {
        for (int i = tid * 16; i < NK * KSTR; i += 256 * 16) *reinterpret_cast<uint4*>(Ks + i) = make_uint4(0, 0, 0, 0);
        for (int i = tid * 16; i < NK * VSTR; i += 256 * 16) *reinterpret_cast<uint4*>(Vs + i) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    const int nrow = min(NK, (Tk + 31) & ~31);                       // sub-tiles that hold at least one key
    for (int idx = tid; idx < nrow * CH; idx += 256) {
        const int row = idx / CH, ch = idx - row * CH;
        const long r = min(row, Tk - 1);
        const uint4 a = *reinterpret_cast<const uint4*>(Kg + r * p.ldk + ch * 8);
        const uint4 c = *reinterpret_cast<const uint4*>(Vg + r * p.ldv + ch * 8);
        *reinterpret_cast<uint4*>(Ks + row * KSTR + ch * 16) = a;
        *reinterpret_cast<uint4*>(Vs + row * VSTR + ch * 16) = c;
    }
    __syncthreads();

    const int tg = lane >> 4, ti = lane & 15;                        // transposed-read lane pattern (see attn_kernel)
    const int tr_row = 4 * (tg >> 1) + (ti >> 2);
    const int tr_col = 16 * (tg & 1) + 4 * (ti & 3);

    for (int it = 0; it < QB; ++it) {
        const int qrow = q_first + it * 128;
        if (qrow - lr >= p.Tq) break;                                // wave-uniform: the wave's 32 rows are past the end
        // ---- S^T = K . Q^T for the sub-tiles that hold keys
        f32x16 sacc[3];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            f32x16 r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (32 * kt < Tk) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + (32 * kt + lr) * KSTR + (16 * ks + 8 * lh) * 2);
                    r = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], r, 0, 0, 0);
                }
            }
            sacc[kt] = r;
        }
        if (it + 1 < QB) load_q(qf, qrow + 128);                     // next block's Q: lands under the softmax / P.V below
        // ---- mask the clamped duplicates, row maximum (in-lane + one cross-half exchange)
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (32 * kt < Tk) {
                if (32 * kt + 32 > Tk) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (32 * kt + (e & 3) + 8 * (e >> 2) + 4 * lh >= Tk) sacc[kt][e] = -1.0e30f;
                }
#pragma unroll
                for (int e = 0; e < 16; e += 2) mx = max3f(mx, sacc[kt][e], sacc[kt][e + 1]);
            }
        }
        mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);
        const float msc = -mx * p.sc;
        f32x2 vsum = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                             // register quad j: keys 32 kt + 8 j + {0..3} + 4 lh
                if (32 * kt + 16 * (j >> 1) < Tk) {                  // (whole 16-key steps: what the P.V MFMAs below consume; masked scores give P = 0)
#pragma unroll
                    for (int e = 4 * j; e < 4 * j + 4; e += 2) {
                        const f32x2 r = exp2_pair(sacc[kt][e], sacc[kt][e + 1], p.sc, msc);
                        sacc[kt][e] = r.x; sacc[kt][e + 1] = r.y;
                        vsum += r;
                    }
                }
            }
        }
        // ---- O^T = V^T . P over the 16-key steps that hold keys
        f32x16 oacc[NDV];
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kb = 32 * kt + 16 * s;
                if (kb < Tk) {
                    f16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (_Float16)sacc[kt][8 * s + j];
#pragma unroll
                    for (int d = 0; d < NDV; ++d) {
                        const unsigned char* a0 = Vs + (kb + tr_row) * VSTR + (32 * d + tr_col) * 2;
                        union { h16x4 h[2]; f16x8 f; } vf;
                        vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)a0);
                        vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(a0 + 8 * VSTR));
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.f, pf, oacc[d], 0, 0, 0);
                    }
                }
            }
        }
        // ---- O = O^T / l, heads merged (as attn_kernel's epilogue)
        const float l = vsum.x + vsum.y;
        const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
        if (qrow < p.Tq) {
            _Float16* og = p.o + (long)b * p.bso + (long)qrow * p.ldo + (long)head * DH;
            auto piece = [&](int d, int eg) __attribute__((always_inline)) {
                const f16x4 h = {(_Float16)(oacc[d][4 * eg + 0] * inv), (_Float16)(oacc[d][4 * eg + 1] * inv),
                                 (_Float16)(oacc[d][4 * eg + 2] * inv), (_Float16)(oacc[d][4 * eg + 3] * inv)};
                return __builtin_bit_cast(u32x2, h);
            };
#pragma unroll
            for (int d = 0; d < NDV; ++d)
#pragma unroll
                for (int eg = 0; eg < 4; eg += 2) {
                    if (p.wide_o && 32 * d + 8 * eg + 16 <= DH) {
                        u32x2 a = piece(d, eg), c = piece(d, eg + 1);
                        const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                        *reinterpret_cast<u32x4*>(og + 32 * d + 8 * eg + 8 * lh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                    } else {
#pragma unroll
                        for (int e2 = eg; e2 < eg + 2; ++e2) {
                            const int dbase = 32 * d + 8 * e2 + 4 * lh;
                            if (dbase < DH) *reinterpret_cast<u32x2*>(og + dbase) = piece(d, e2);
                        }
                    }
                }
        }
    }
}


#ifdef MLSD_GEMM_EXPERIMENTS   /* built, correct and SLOWER than the tile-loop kernels (profiles/NOTES.md "ping-pong attention"): kept reproducible, not in the product build */
// ---------------------------------------------------------------------------------------------------------------------
// d_head = 64, no causal mask: the PING-PONG kernel.
//   PMC on the kernels above: matrix pipes busy 28-39 % + vector ALU ~55 % ~ 94 %: a wave's QK^T -> softmax -> P.V chain runs
//   almost serially and the 2-3 waves of a SIMD overlap little (compiler-ordered software pipelining lost to occupancy twice,
//   DESIGN.md section 9.2).  Here the overlap is built into the block structure instead of the instruction order, the way the
//   ping-pong GEMM tiles do it (gemm_pp.hpp):
//     (operands of one (batch, head) must span < 2 GiB: 32-bit buffer offsets; the launcher checks)
//     block = 8 waves = group A (waves 0-3) + group B (waves 4-7), one wave of each group per SIMD, NB x 32 query rows per wave;
//     the block alternates PHASES separated by ONE s_barrier:      phase g:  A: matrix   B: vector     g+1:  A: vector   B: matrix
//       matrix phase (tile t):  O^T += V^T(t-1) . P(t-1)   then   S'(t) = K(t) . Q'^T       16 NB MFMAs, nothing else but LDS reads
//       vector phase (tile t):  row max, (rare) rescale, P = exp2(S'), row sums                        no MFMA, no LDS
//     so one SIMD always has one wave on its matrix pipe and the other on its vector ALU.  B runs one phase behind A on the SAME
//     K / V tiles: K(t+1) and V(t) are staged (LDS-DMA, issued by group A at the start of its matrix phase t, two-slot rings)
//     into the slots whose last reader (B, phase before) has passed the barrier, and are first read two phases later.
//   The softmax argument costs no vector instruction: Q' = Q * (log2(e) / sqrt(d)) is folded into the query fragments once per
//   block (fp32 product, one rounding), and the S' accumulators START at -m (the running, deferred maximum in the exp2 domain,
//   known before the tile's QK^T because the vector phase of the previous tile has finished), so P = exp2(acc) directly;
//   the 16 NB x 2 accumulator initialisations ride in the matrix phase, where the vector ALU idles.
// PRIO: s_setprio 1 around 0 = nothing, 1 = the MFMA clusters, 2 = the vector phase.  (MI355X_MICROARCH.md "Two waves per SIMD" item 2:
// with the matrix wave at priority 1 its partner's vector instructions only get the left-over issue slots.)
template <int NB, int WPS, int PRIO>      // NB: 32-row query blocks per wave; WPS: waves per SIMD the register allocation aims for (2 per resident block)
__global__ __launch_bounds__(512, WPS) void attn64pp_kernel(const AttnP p)
{
    constexpr int DH = 64, RB = 128;                 // bytes per K/V row
    __shared__ __attribute__((aligned(1024))) unsigned char Ks2[2][64 * RB];
    __shared__ __attribute__((aligned(1024))) unsigned char Vs2[2][64 * RB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;        // group (0 = A, 1 = B), wave inside the group
    const int lr = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gk = slot / p.nq, qb = slot - gk * p.nq;
    const int g_ = gk * 8 + xcd;
    if (g_ >= p.G) return;
    const int head = g_ % p.n_head, b = g_ / p.n_head;
    const int qw = qb * (256 * NB) + wave * (32 * NB);       // first query row of this wave

    const _Float16* Qg = p.q + (long)b * p.bsq + (long)head * DH;
    const _Float16* Kg = p.k + (long)b * p.bsk + (long)head * DH;
    const _Float16* Vg = p.v + (long)b * p.bsv + (long)head * DH;

    // Q' fragments (B operand of S'^T = K . Q'^T): lane (lr = query, lh) holds Q[q][16 ks + 8 lh + j] * sc, rounded once
    f16x8 qf[NB][4];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 raw = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Qg + (long)(qw + 32 * sb + lr) * p.ldq + 16 * ks + 8 * lh));
            f16x8 sc8;
#pragma unroll
            for (int j = 0; j < 8; ++j) sc8[j] = (_Float16)((float)raw[j] * p.sc);
            qf[sb][ks] = sc8;
        }

    const int nt = (p.Tk + 63) / 64;
    // LDS-DMA staging by the 4 waves of group A: wave w fills rows 16w .. 16w+15 of a tile with two 1-KiB pieces per operand
    // (K image: slot = chunk ^ ((row >> 1) & 7), V image: slot = chunk ^ (((row >> 1) & 1) << 2): see attn64x2_kernel)
    // buffer_load ... lds through one descriptor per operand (base = this (batch, head)'s first row, range = its Tk rows): the
    // per-lane part of the address is ONE 32-bit offset (row inside the wave's 8-row piece + swizzled chunk), the tile / piece
    // advance is a scalar offset, and rows past Tk fail the range check and arrive as ZEROS (their scores are masked, P = 0).
    // The lane offsets are RECOMPUTED at every use from an opaque copy of the lane id: kept in registers across the tile loop
    // they were spilled, and every reload carried a compiler-inserted s_waitcnt vmcnt(0) in front of the DMA it fed -- each piece
    // then waited for the previous one to land (the pitfall of cdna_hip_programming.md, "lane-constant address hoisted ...").
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, (int)(((long)(p.Tk - 1) * p.ldk + DH) * 2), 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, (int)(((long)(p.Tk - 1) * p.ldv + DH) * 2), 0x00020000);
    auto stage_k = [&](int t) __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));      // lane id, never live across the loop
        const int srow = l >> 3, sslot = l & 7;
        const int kvo0 = srow * (int)p.ldk * 2 + ((sslot ^ (srow >> 1)) << 4);              // piece 0 of a wave: rows 16 wq + srow
        const int kvo1 = kvo0 ^ 64;                                                         // piece 1 (rows + 8): swizzle term + 4
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(Ks2[t & 1] + (wq * 16 + i * 8) * RB), 16,
                                                     i ? kvo1 : kvo0, (t * 64 + wq * 16 + i * 8) * (int)p.ldk * 2, 0, 0);
    };
    auto stage_v = [&](int t) __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int srow = l >> 3, sslot = l & 7;
        const int vvo = srow * (int)p.ldv * 2 + ((sslot ^ (((srow >> 1) & 1) << 2)) << 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(Vs2[t & 1] + (wq * 16 + i * 8) * RB), 16,
                                                     vvo, (t * 64 + wq * 16 + i * 8) * (int)p.ldv * 2, 0, 0);
    };

    f32x16 oacc[NB][2];                               // O^T accumulators: [query block][d block of 32]
    f32x16 sacc[NB][2];                               // S'^T accumulators: [query block][32-key sub-tile]
    float msc[NB];                                    // running (deferred) maximum of the scaled scores, exp2 domain
    f32x2 vsum[NB];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
        msc[sb] = 0.f; vsum[sb] = f32x2{0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) { oacc[sb][d][e] = 0.f; sacc[sb][d][e] = 0.f; }
    }

    if (grp == 0) stage_k(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K(0) (group A's pieces) and everybody's Q
    __syncthreads();

    // fragment addressing (as attn64x2_kernel)
    const int kswz = (lr >> 1) & 7;
    const int tg = lane >> 4, ti = lane & 15;
    const int tr_row = 4 * (tg >> 1) + (ti >> 2);
    const int vswz = ((tr_row >> 1) & 1) << 2;
    const int tr_c = 2 * (tg & 1) + ((ti & 3) >> 1), tr_b = 8 * (ti & 1);

    // Phase bodies.  qk(t): S'(t) = -m + K(t) . Q'^T (group A also issues the staging of K(t+1) and V(t): both targets were last
    // read, by group B, in the phase that just ended).  pv(t): O^T += V^T(t) . P(t).  sm(t): the softmax of tile t.
    // Matrix phase, NB = 1 (register room for it): ALL LDS reads of the phase -- the 16 transposed V reads of P.V and the 8 row reads
    // of the next QK^T, 64 registers -- are issued up front, then the 16 MFMAs run against counted lgkmcnt waits.  With one wave per
    // SIMD on the matrix pipe nobody hides a read -> wait -> MFMA chain: issued group by group the phase took twice its MFMA time
    // (timing-only builds, tools/attn_bench.py: matrix phases alone 321 us of a 393 us launch at 4096 x 4096).
    auto stage_next = [&](int t) __attribute__((always_inline)) {     // group A: K(t+1) and V(t) into the slots group B finished with a phase ago
        if (grp == 0) {
            if (t + 1 < nt) stage_k(t + 1);
            stage_v(t);
        }
    };
    auto init_s = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) {
            const float ini = -msc[sb];
#pragma unroll
            for (int e = 0; e < 16; ++e) { sacc[sb][0][e] = ini; sacc[sb][1][e] = ini; }
        }
    };
    auto read_k = [&](int t, f16x8 (&kf)[8]) __attribute__((always_inline)) {
        const unsigned char* Ks = Ks2[t & 1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((2 * ks + lh) ^ kswz) << 4;
            kf[2 * ks] = *reinterpret_cast<const f16x8*>(Ks + lr * RB + off);
            kf[2 * ks + 1] = *reinterpret_cast<const f16x8*>(Ks + (32 + lr) * RB + off);
        }
    };
    auto mfma_qk = [&](const f16x8 (&kf)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) {
                sacc[sb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2 * ks], qf[sb][ks], sacc[sb][0], 0, 0, 0);
                sacc[sb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2 * ks + 1], qf[sb][ks], sacc[sb][1], 0, 0, 0);
            }
    };
    auto qk = [&](int t) __attribute__((always_inline)) {
        stage_next(t);
        f16x8 kf[8];
        read_k(t, kf);
        init_s();
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        mfma_qk(kf);
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };
    auto pv = [&](int t) __attribute__((always_inline)) {
        const unsigned char* Vs = Vs2[t & 1];
        const bool sub1 = t * 64 + 32 < p.Tk;        // the tile's second 32-key sub-tile holds at least one key (wave-uniform)
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1 && !sub1) continue;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const int kb = 32 * kt + 16 * sx;
                f16x8 pf[NB];                         // P (fp32, left in the S' accumulators by sm) -> fp16 B operand, permuted k order
#pragma unroll
                for (int sb = 0; sb < NB; ++sb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[sb][j] = (_Float16)sacc[sb][kt][8 * sx + j];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* a0 = Vs + (kb + tr_row) * RB + (((4 * d + tr_c) ^ vswz) << 4) + tr_b;
                    union { h16x4 h[2]; f16x8 f; } vf;
                    vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)a0);
                    vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(a0 + 8 * RB));
#pragma unroll
                    for (int sb = 0; sb < NB; ++sb)
                        oacc[sb][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.f, pf[sb], oacc[sb][d], 0, 0, 0);
                }
            }
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };
    // pv(t) + qk(t+1) as ONE phase with every LDS read in front (NB = 1; whole tiles only: a ragged last tile takes pv / qk)
    auto pv_qk = [&](int t) __attribute__((always_inline)) {
        stage_next(t + 1);
        const unsigned char* Vs = Vs2[t & 1];
        union VF { h16x4 h[2]; f16x8 f; };
        VF vf[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {                 // g = (kt, sx, d)
            const int kb = 16 * (g >> 1), d = g & 1;
            const unsigned char* a0 = Vs + (kb + tr_row) * RB + (((4 * d + tr_c) ^ vswz) << 4) + tr_b;
            vf[g].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)a0);
            vf[g].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)(a0 + 8 * RB));
        }
        f16x8 kf[8];
        read_k(t + 1, kf);
        f16x8 pf[NB][4];
#pragma unroll
        for (int sb = 0; sb < NB; ++sb)
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[sb][h][j] = (_Float16)sacc[sb][h >> 1][8 * (h & 1) + j];
        __builtin_amdgcn_sched_barrier(0);            // reads and conversions above, MFMAs below (the compiler counts the lgkmcnt waits)
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int sb = 0; sb < NB; ++sb)
                oacc[sb][g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[g].f, pf[sb][g >> 1], oacc[sb][g & 1], 0, 0, 0);
        init_s();
        mfma_qk(kf);
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    };
    auto sm = [&](int t) __attribute__((always_inline)) {
        const int kv0 = t * 64;
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) {
            f32x16& sa = sacc[sb][0];
            f32x16& sc_ = sacc[sb][1];
            if (kv0 + 64 > p.Tk) {                    // ragged last tile (wave-uniform): mask the clamped duplicates
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (key >= p.Tk) sa[e] = -1.0e30f;
                    if (key + 32 >= p.Tk) sc_[e] = -1.0e30f;
                }
            }
            float mx = max3f(sa[0], sa[1], sc_[0]);
            mx = max3f(mx, sc_[1], sa[2]);
#pragma unroll
            for (int e = 3; e < 16; e += 2) mx = max3f(mx, sa[e], (e + 1 < 16) ? sa[e + 1] : sa[e]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = max3f(mx, sc_[e], sc_[e + 1]);
            mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);
            // deferred rescale (T13): the reference maximum moves only when a row's tile maximum exceeds it by more than 2^6
            // (P <= 64 keeps full relative precision in fp16); always in the first tile (m starts at 0, not at a maximum).
            // The decision covers the whole wave and is taken BEFORE this tile's P is formed: O, l and P share one reference.
            if (t == 0 || __any(mx > 6.0f)) {
                const float dlt = t == 0 ? mx : fmaxf(mx, 0.f);
                msc[sb] += dlt;
                if (t > 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-dlt);
                    vsum[sb] *= alpha;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oacc[sb][d][e] *= alpha;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) { sa[e] -= dlt; sc_[e] -= dlt; }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16& sx_ = kt ? sc_ : sa;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 r = {__builtin_amdgcn_exp2f(sx_[e]), __builtin_amdgcn_exp2f(sx_[e + 1])};
                    sx_[e] = r.x; sx_[e + 1] = r.y;
                    vsum[sb] += r;
                }
            }
        }
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K(t+1), V(t) issued a phase ago have landed
    };
    auto phase_end = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                 // raw barrier (no vmcnt drain): LDS-DMA pieces stay in flight across it
        __builtin_amdgcn_sched_barrier(0);
    };
    // Global phases g = 0 .. 2 nt + 1, one barrier after each of g = 0 .. 2 nt (2 nt + 1 barriers for every wave):
    //   group A:  g = 0: qk(0)   g = 2t+1: sm(t)   g = 2t+2: pv(t) [+ qk(t+1)]          (ends with pv(nt-1) at g = 2 nt, then its epilogue)
    //   group B:  the same sequence one phase later (idle at g = 0, pv(nt-1) at g = 2 nt + 1 with no barrier behind it)
    const int dbg = p.causal;                         // timing-only builds of the loop (tools/attn_bench.py): 1 = no vector phase, 2 = no matrix phase
    auto stamp = [&](int t, int k) __attribute__((always_inline)) {      // diagnostics: s_memtime at the phase boundaries of block 0
        if (p.tbuf && blockIdx.x == 0 && t < 16 && lane == 0) p.tbuf[(wave * 16 + t) * 5 + k] = __builtin_readcyclecounter();
    };
    if (grp == 1) phase_end();
    qk(0);
    phase_end();
    for (int t = 0; t < nt; ++t) {
        stamp(t, 0);
        if (!(dbg & 1)) sm(t);
        stamp(t, 1);
        phase_end();
        stamp(t, 2);
        if (NB == 1 && WPS == 2 && t + 1 < nt && (t + 1) * 64 <= p.Tk && !(dbg & 2)) pv_qk(t);
        else {
            if (!(dbg & 2)) pv(t);
            __builtin_amdgcn_sched_barrier(0);        // P(t) is dead here: the accumulators restart at -m for the next tile
            if (t + 1 < nt && !(dbg & 2)) qk(t + 1);
        }
        stamp(t, 3);
        if (t + 1 < nt || grp == 0) phase_end();
        stamp(t, 4);
    }

    // ---- epilogue: O = O^T / l, heads merged.  o[d][e]: d-index = 32*d + (e&3) + 8*(e>>2) + 4*lh, query = lane&31
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
        const float l = vsum[sb].x + vsum[sb].y;      // a lane half holds the keys 4*lh + {0..3} + 8j of every 32
        const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
        _Float16* og = p.o + (long)b * p.bso + (long)(qw + 32 * sb + lr) * p.ldo + (long)head * DH;
        auto piece = [&](int d, int eg) __attribute__((always_inline)) {
            const f16x4 h = {(_Float16)(oacc[sb][d][4 * eg + 0] * inv), (_Float16)(oacc[sb][d][4 * eg + 1] * inv),
                             (_Float16)(oacc[sb][d][4 * eg + 2] * inv), (_Float16)(oacc[sb][d][4 * eg + 3] * inv)};
            return __builtin_bit_cast(u32x2, h);
        };
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int eg = 0; eg < 4; eg += 2) {
                if (p.wide_o) {                       // 16 bytes per lane (see attn_kernel's epilogue)
                    u32x2 a = piece(d, eg), c = piece(d, eg + 1);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
                    *reinterpret_cast<u32x4*>(og + 32 * d + 8 * eg + 8 * lh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                } else {
                    *reinterpret_cast<u32x2*>(og + 32 * d + 8 * eg + 4 * lh) = piece(d, eg);
                    *reinterpret_cast<u32x2*>(og + 32 * d + 8 * (eg + 1) + 4 * lh) = piece(d, eg + 1);
                }
            }
    }
}
#endif  // MLSD_GEMM_EXPERIMENTS

int g_attn_force_old = 0;   // diagnostics / A-B timing: 1 = never use attn64x2_kernel
// the 256-row blocks quantise badly on short sequences (Tq 1024 x 160 groups = 640 blocks on 512 slots: measured slower than
// the general kernel), so they take Tq >= 2048 only
int g_attn_x2_min_tq = 2048;
int g_attn_sp = -1;         // (-1: MLSD_ATTN_SP from the environment, default 1)  d_head 64 launches with whole key tiles (Tk % 64 == 0, >= 128) and Tq % 256 == 0, Tq >= 768 and at least 128 blocks run software-pipelined inside the wave (attn64x2s_kernel; 0 = the tile-loop kernels, 2 = from Tq = 256 on: A/B and kernel tests, mlsd_attention_sp)
int g_attn_wide_o = 1;      // 16-byte output stores (0 = 8-byte pieces; A/B timing)
#ifdef MLSD_GEMM_EXPERIMENTS
int g_attn_pp = 0;          // ping-pong kernel for d_head 64 (measured: parity with the tile-loop kernels; off by default): 0 = off, 1 = by shape (Tq % 512 == 0 and >= 2048: 64 rows per wave; Tq % 256 == 0: 32, two blocks per CU), 2 = always 32 rows, 3 = always 64 rows, 4 = 32 rows, one block per CU
#endif
int g_attn_tk96 = 1;        // Tk <= 96 without a causal mask: the one-pass kernel (0 = the general kernels; A/B timing)
int g_attn_tk96_qb = 0;     // query blocks of 128 rows per workgroup (0 = by the launch size)
int g_attn_vsum = 1;        // row sums on the VALU (v_pk_add_f32) instead of ones.P MFMAs: +4..7 % on the SDXL shapes (tools/attn_bench.py); 0 = matrix-pipe sums

bool attn_sp_takes(const mlsd_attn_args* a)
{
    if (g_attn_sp < 0) { const char* e = getenv("MLSD_ATTN_SP"); g_attn_sp = (e && *e >= '0' && *e <= '2') ? *e - '0' : 1; }
    return g_attn_sp && (a->d_head == 64 || a->d_head == 40) && !a->causal && !(a->Tq & 255) && (g_attn_sp == 2 ? a->Tq >= 256 : (a->Tq >= 768 && (long)a->n_head * a->n_batch * (a->Tq / 256) >= 128)) &&      /* (default: from 768 tokens on and at least 128 blocks of 256 rows -- below that the 128-row blocks of the tile-loop kernel fill the chip better: tools/attn_sp_small_shapes.py) */ !(a->Tk & 63) && a->Tk >= 128 &&
           (long)a->Tk * a->ldk < (1L << 30) && (long)a->Tk * a->ldv < (1L << 30) && !(((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15) && !((a->ldq | a->ldk | a->ldv) & 7);
}

int launch_attn64x2(const mlsd_attn_args* a, hipStream_t st)
{
    AttnP p; p.tbuf = nullptr;
    p.q = (const _Float16*)a->q; p.k = (const _Float16*)a->k; p.v = (const _Float16*)a->v; p.o = (_Float16*)a->out;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.n_head = a->n_head; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = 0;
    p.sc = (float)(1.4426950408889634 / sqrt(64.0));
    p.nq = a->Tq / 256; p.G = a->n_head * a->n_batch;
    p.wide_o = g_attn_wide_o && !(a->ldo & 7) && !(a->bso & 7) && !((uintptr_t)a->out & 15);
    const dim3 grid((unsigned)(8 * ((p.G + 7) / 8) * p.nq));
    if (attn_sp_takes(a)) {      // whole key tiles: the software-pipelined form
        if (a->d_head == 40) { p.sc = (float)(1.4426950408889634 / sqrt(40.0)); hipLaunchKernelGGL(attn64x2s_kernel<40>, grid, dim3(256), 0, st, p); }
        else hipLaunchKernelGGL(attn64x2s_kernel<64>, grid, dim3(256), 0, st, p);
        return mlsd_check_launch("attn64x2s_kernel");
    }
    if (g_attn_vsum) hipLaunchKernelGGL(attn64x2_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(attn64x2_kernel<false>, grid, dim3(256), 0, st, p);
    return mlsd_check_launch("attn64x2_kernel");
}

template <int DH>
int launch_attn(const mlsd_attn_args* a, hipStream_t st)
{
    AttnP p; p.tbuf = nullptr;
    p.q = (const _Float16*)a->q; p.k = (const _Float16*)a->k; p.v = (const _Float16*)a->v; p.o = (_Float16*)a->out;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.n_head = a->n_head; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = a->causal;
    p.sc = (float)(1.4426950408889634 / sqrt((double)a->d_head));
    p.nq = (a->Tq + 127) / 128; p.G = a->n_head * a->n_batch;
    p.wide_o = g_attn_wide_o && !(a->ldo & 7) && !(a->bso & 7) && !((uintptr_t)a->out & 15);
    const dim3 grid((unsigned)(8 * ((p.G + 7) / 8) * p.nq));
    if (g_attn_vsum) hipLaunchKernelGGL((attn_kernel<DH, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_kernel<DH, false>), grid, dim3(256), 0, st, p);
    return mlsd_check_launch("attn_kernel");
}


unsigned long long* g_attn_tbuf = nullptr;
#ifdef MLSD_GEMM_EXPERIMENTS
int g_attn_pp_prio = 0, g_attn_pp_dbg = 0;
template <int NB, int WPS>
int launch_attn64pp(const mlsd_attn_args* a, hipStream_t st)
{
    AttnP p; p.tbuf = nullptr;
    p.q = (const _Float16*)a->q; p.k = (const _Float16*)a->k; p.v = (const _Float16*)a->v; p.o = (_Float16*)a->out;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.n_head = a->n_head; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = g_attn_pp_dbg; p.tbuf = g_attn_tbuf;
    p.sc = (float)(1.4426950408889634 / sqrt(64.0));
    p.nq = a->Tq / (256 * NB); p.G = a->n_head * a->n_batch;
    p.wide_o = g_attn_wide_o && !(a->ldo & 7) && !(a->bso & 7) && !((uintptr_t)a->out & 15);
    const dim3 grid((unsigned)(8 * ((p.G + 7) / 8) * p.nq));
    if (g_attn_pp_prio == 1) hipLaunchKernelGGL((attn64pp_kernel<NB, WPS, 1>), grid, dim3(512), 0, st, p);
    else if (g_attn_pp_prio == 2) hipLaunchKernelGGL((attn64pp_kernel<NB, WPS, 2>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((attn64pp_kernel<NB, WPS, 0>), grid, dim3(512), 0, st, p);
    return mlsd_check_launch("attn64pp_kernel");
}
#endif

template <int DH>
int launch_attn_tk96(const mlsd_attn_args* a, hipStream_t st)
{
    AttnP p; p.tbuf = nullptr;
    p.q = (const _Float16*)a->q; p.k = (const _Float16*)a->k; p.v = (const _Float16*)a->v; p.o = (_Float16*)a->out;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.n_head = a->n_head; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = 0;
    p.sc = (float)(1.4426950408889634 / sqrt((double)a->d_head));
    p.G = a->n_head * a->n_batch;
    p.wide_o = g_attn_wide_o && !(a->ldo & 7) && !(a->bso & 7) && !((uintptr_t)a->out & 15);
    const int G8 = 8 * ((p.G + 7) / 8), nblk128 = (a->Tq + 127) / 128;
    // query blocks per workgroup: the keys are staged once per workgroup, so fewer, longer workgroups stage less -- as long as
    // the launch still holds >= 2 workgroups per CU to hide the Q / O latency of one behind the arithmetic of another
    int QB = g_attn_tk96_qb;
    if (QB <= 0) { QB = 8; while (QB > 1 && (long)G8 * ((nblk128 + QB - 1) / QB) < 512) QB >>= 1; }
    p.nq = (nblk128 + QB - 1) / QB;
    const dim3 grid((unsigned)(G8 * p.nq));
    if (DH == 64 && g_attn_tk96 == 2) hipLaunchKernelGGL((attn_tk96_kernel<DH, (DH == 64 ? 4 : 2)>), grid, dim3(256), 0, st, p, QB);   // (A/B: 127 registers + 2 spilled)
    else hipLaunchKernelGGL((attn_tk96_kernel<DH, (DH <= 80 ? 3 : 2)>), grid, dim3(256), 0, st, p, QB);
    return mlsd_check_launch("attn_tk96_kernel");
}

}  // namespace

extern "C" {

MLSD_API int mlsd_attention(const mlsd_attn_args* a, void* stream)
{
    if (!a || !a->q || !a->k || !a->v || !a->out) return mlsd_set_error(-1, "mlsd_attention: null operand");
    if (a->Tq <= 0 || a->Tk <= 0 || a->n_batch <= 0 || a->n_head <= 0) return mlsd_set_error(-1, "mlsd_attention: empty problem");
    if ((a->ldq & 7) || (a->ldk & 7) || (a->ldv & 7) || (a->ldo & 3)) return mlsd_set_error(-1, "mlsd_attention: strides must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    if (g_attn_tk96 && !a->causal && a->Tk <= 96) {
        switch (a->d_head) {
        case 40: return launch_attn_tk96<40>(a, st);
        case 64: return launch_attn_tk96<64>(a, st);
        case 80: return launch_attn_tk96<80>(a, st);
        case 160: return launch_attn_tk96<160>(a, st);
        default: break;
        }
    }
    switch (a->d_head) {
    case 32: return launch_attn<32>(a, st);
    case 40:
        if (!g_attn_force_old && attn_sp_takes(a)) return launch_attn64x2(a, st);      // round 6: SD1.5's 4096-token level on the software-pipelined kernel, run as d = 64
        return launch_attn<40>(a, st);
    case 64:
#ifdef MLSD_GEMM_EXPERIMENTS
        if (g_attn_pp && !g_attn_force_old && !a->causal && !(a->Tq & 255) && !(((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15) &&
            (long)a->Tk * a->ldk < (1L << 30) && (long)a->Tk * a->ldv < (1L << 30)) {
            const bool big = !(a->Tq & 511) && a->Tq >= 2048;
            if (g_attn_pp == 3 ? !(a->Tq & 511) : (g_attn_pp == 1 && big)) return launch_attn64pp<2, 2>(a, st);
            if (g_attn_pp == 4) return launch_attn64pp<1, 2>(a, st);
            return launch_attn64pp<1, 4>(a, st);
        }
#endif
        // every q/k/v row must be 16-byte aligned for the LDS-DMA pieces (strides are multiples of 8 halfs: checked above)
        if (!g_attn_force_old && !a->causal && (a->Tq >= g_attn_x2_min_tq || attn_sp_takes(a)) && !(a->Tq & 255) &&
            !(((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15)) return launch_attn64x2(a, st);
        return launch_attn<64>(a, st);
    case 80: return launch_attn<80>(a, st);
    case 160: return launch_attn<160>(a, st);
    default: return mlsd_set_error(-1, "mlsd_attention: unsupported d_head %d (supported: 32,40,64,80,160)", a->d_head);
    }
}

MLSD_API void mlsd_attention_force_old(int on) { g_attn_force_old = on; }
MLSD_API void mlsd_attention_x2_min_tq(int tq) { g_attn_x2_min_tq = tq; }
MLSD_API void mlsd_attention_sp(int mode) { g_attn_sp = (mode >= 0 && mode <= 2) ? mode : -1; }
MLSD_API void mlsd_attention_vsum(int on) { g_attn_vsum = on; }
MLSD_API void mlsd_attention_wide_stores(int on) { g_attn_wide_o = on; }
#ifdef MLSD_GEMM_EXPERIMENTS
MLSD_API void mlsd_attention_pp(int mode) { g_attn_pp = mode & 15; g_attn_pp_prio = (mode >> 4) & 3; g_attn_pp_dbg = (mode >> 8) & 3; }
#else
MLSD_API void mlsd_attention_pp(int mode) { (void)mode; }       /* the ping-pong attention kernel is not in the product build (make EXPERIMENTS=1) */
#endif
MLSD_API void mlsd_attention_set_trace(void* buf) { g_attn_tbuf = (unsigned long long*)buf; }   /* 8 waves x 16 tiles x 5 stamps of block 0 */
MLSD_API void mlsd_attention_tk96(int on, int qb) { g_attn_tk96 = on; g_attn_tk96_qb = qb; }

}  // extern "C"
