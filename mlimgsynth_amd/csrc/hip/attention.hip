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

namespace {

struct AttnP {
    const _Float16 *q, *k, *v;
    _Float16* o;
    long ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    int n_head, Tq, Tk, causal;
    float sc;  // 1/sqrt(d_head) * log2(e)
};

template <int DH>
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
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128, qw = q0 + wave * 32;
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
    float m_run = -1.0e30f, l_run = 0.f;

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
        float mx = -1.0e30f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // deferred rescale (T13): keep the old reference maximum while the new one exceeds it by less than
        // 2^6 in the exp2 domain (P <= 64 fits fp16 with full relative precision); the decision is taken for
        // the whole wave, BEFORE this tile's P is formed, so O, l and P always share one reference.
        const bool grow = (mx - m_run) * p.sc > 6.0f;
        if (__any(grow)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.sc);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < NDV; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }
        const float msc = -m_run * p.sc;
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kt][e], p.sc, msc));   // raw v_exp_f32: arguments <= 6, underflow to 0 is the wanted result
                sacc[kt][e] = pv;
                rs += pv;
            }
        l_run += rs;

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
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Tq) {
        _Float16* og = p.o + (long)b * p.bso + (long)qrow * p.ldo + (long)head * DH;
#pragma unroll
        for (int d = 0; d < NDV; ++d)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                const int dbase = 32 * d + 8 * eg + 4 * lh;
                if (dbase < DH) {
                    f16x4 h = {(_Float16)(oacc[d][4 * eg + 0] * inv), (_Float16)(oacc[d][4 * eg + 1] * inv),
                               (_Float16)(oacc[d][4 * eg + 2] * inv), (_Float16)(oacc[d][4 * eg + 3] * inv)};
                    *reinterpret_cast<f16x4*>(og + dbase) = h;
                }
            }
    }
}

template <int DH>
int launch_attn(const mlsd_attn_args* a, hipStream_t st)
{
    AttnP p;
    p.q = (const _Float16*)a->q; p.k = (const _Float16*)a->k; p.v = (const _Float16*)a->v; p.o = (_Float16*)a->out;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.n_head = a->n_head; p.Tq = a->Tq; p.Tk = a->Tk; p.causal = a->causal;
    p.sc = (float)(1.4426950408889634 / sqrt((double)a->d_head));
    const dim3 grid((a->Tq + 127) / 128, a->n_head, a->n_batch);
    hipLaunchKernelGGL(attn_kernel<DH>, grid, dim3(256), 0, st, p);
    return mlsd_check_launch("attn_kernel");
}

}  // namespace

extern "C" {

MLSD_API int mlsd_attention(const mlsd_attn_args* a, void* stream)
{
    if (!a || !a->q || !a->k || !a->v || !a->out) return mlsd_set_error(-1, "mlsd_attention: null operand");
    if (a->Tq <= 0 || a->Tk <= 0 || a->n_batch <= 0 || a->n_head <= 0) return mlsd_set_error(-1, "mlsd_attention: empty problem");
    if ((a->ldq & 7) || (a->ldk & 7) || (a->ldv & 7) || (a->ldo & 3)) return mlsd_set_error(-1, "mlsd_attention: strides must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    switch (a->d_head) {
    case 32: return launch_attn<32>(a, st);
    case 40: return launch_attn<40>(a, st);
    case 64: return launch_attn<64>(a, st);
    case 80: return launch_attn<80>(a, st);
    case 160: return launch_attn<160>(a, st);
    default: return mlsd_set_error(-1, "mlsd_attention: unsupported d_head %d (supported: 32,40,64,80,160)", a->d_head);
    }
}

}  // extern "C"
