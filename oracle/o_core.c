/* ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Core: tensors, fp16 rounding, blocked SGEMM (OpenMP + AVX2/FMA), parameter store
 * with the deterministic synthetic-weight generator.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <immintrin.h>
#include <omp.h>

/* ------------------------------------------------------------------ tensors */
/* ------------------------------------------------------------------ block cache
 * Every op output, every rounded operand copy and the SGEMM's panel buffers used to be a fresh posix_memalign / free pair: above glibc's mmap threshold that is an mmap, first-touch
 * page faults taken by ALL OpenMP threads at once inside the next parallel loop, and an munmap with its TLB shoot-down -- on the GPU box's 2 x 64-core host this, not arithmetic, was
 * the CPU baseline's time (round 5: the new SGEMM ran at 7.4 TFLOP/s stand-alone on 64 threads and the SD1.5 evaluation still took 4.6 s, 87 % of it inside orc_sgemm_nt calls).
 * Blocks of >= 256 KB are kept on a small free list and handed out again (best fit within 25 %); the capacity sits in a 64-byte header in front of the data. */
#include <pthread.h>
#define OBC_MIN ((size_t)256 << 10)
#define OBC_SLOTS 96
#define OBC_MAX_BYTES ((size_t)8 << 30)
static struct { void* p; size_t cap; } g_obc[OBC_SLOTS];
static size_t g_obc_bytes = 0;
static pthread_mutex_t g_obc_mu = PTHREAD_MUTEX_INITIALIZER;

void* orc_balloc(size_t bytes)
{
	if (bytes < OBC_MIN) { void *p = NULL; if (posix_memalign(&p, 64, bytes + 64)) return NULL; ((size_t*)p)[0] = 0; return (char*)p + 64; }
	void *hit = NULL; size_t hcap = 0;
	pthread_mutex_lock(&g_obc_mu);
	int best = -1;
	for (int i=0;i<OBC_SLOTS;++i) if (g_obc[i].p && g_obc[i].cap >= bytes && g_obc[i].cap <= bytes + bytes / 4 && (best < 0 || g_obc[i].cap < g_obc[best].cap)) best = i;
	if (best >= 0) { hit = g_obc[best].p; hcap = g_obc[best].cap; g_obc[best].p = NULL; g_obc_bytes -= hcap; }
	pthread_mutex_unlock(&g_obc_mu);
	if (hit) { ((size_t*)hit)[0] = hcap; return (char*)hit + 64; }
	void *p = NULL;
	if (posix_memalign(&p, 64, bytes + 64)) return NULL;
	((size_t*)p)[0] = bytes;
	return (char*)p + 64;
}

void orc_bfree(void* d)
{
	if (!d) return;
	void *p = (char*)d - 64;
	const size_t cap = ((size_t*)p)[0];
	if (cap >= OBC_MIN) {
		pthread_mutex_lock(&g_obc_mu);
		int slot = -1;
		if (g_obc_bytes + cap <= OBC_MAX_BYTES) for (int i=0;i<OBC_SLOTS;++i) if (!g_obc[i].p) { slot = i; break; }
		if (slot >= 0) { g_obc[slot].p = p; g_obc[slot].cap = cap; g_obc_bytes += cap; }
		pthread_mutex_unlock(&g_obc_mu);
		if (slot >= 0) return;
	}
	free(p);
}

void orc_bcache_drop(void)      /* returns the cached blocks to the system */
{
	pthread_mutex_lock(&g_obc_mu);
	for (int i=0;i<OBC_SLOTS;++i) if (g_obc[i].p) { free(g_obc[i].p); g_obc[i].p = NULL; }
	g_obc_bytes = 0;
	pthread_mutex_unlock(&g_obc_mu);
}

OT* ot_new(int64_t n0, int64_t n1, int64_t n2, int64_t n3)
{
	OT *t = (OT*)calloc(1, sizeof(OT));
	t->ne[0]=n0; t->ne[1]=n1; t->ne[2]=n2; t->ne[3]=n3;
	int64_t n = n0*n1*n2*n3;
	t->d = (float*)orc_balloc((size_t)(n>0?n:1)*sizeof(float));
	if (!t->d) { free(t); return NULL; }
	return t;
}

OT* ot_from(const float* src, int64_t n0, int64_t n1, int64_t n2, int64_t n3)
{
	OT *t = ot_new(n0,n1,n2,n3);
	memcpy(t->d, src, (size_t)ot_nel(t)*sizeof(float));
	return t;
}

void ot_free(OT* t) { if (t) { orc_bfree(t->d); free(t); } }

int64_t ot_nel(const OT* t) { return t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]; }

static int g_threads = 0;
void orc_set_threads(int n) { g_threads = n; if (n > 0) omp_set_num_threads(n); }
int  orc_get_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

/* round through binary16, RNE (what ggml_fp32_to_fp16_row / GGML_FP32_TO_FP16 do) */
void orc_round_f16(float* x, int64_t n)
{
	#pragma omp parallel for schedule(static) if (n > 65536)
	for (int64_t i=0; i<n-7; i+=8) {
		__m256 v = _mm256_loadu_ps(x+i);
		__m128i h = _mm256_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT|_MM_FROUND_NO_EXC);
		_mm256_storeu_ps(x+i, _mm256_cvtph_ps(h));
	}
	for (int64_t i=n&~(int64_t)7; i<n; ++i)
		x[i] = _cvtsh_ss(_cvtss_sh(x[i], _MM_FROUND_TO_NEAREST_INT|_MM_FROUND_NO_EXC));
}

/* dst = src rounded through binary16 (one parallel pass: a fresh destination's pages are first touched by all threads) */
void orc_round_f16_copy(float* dst, const float* src, int64_t n)
{
	#pragma omp parallel for schedule(static) if (n > 65536)
	for (int64_t i=0; i<n-7; i+=8)
		_mm256_storeu_ps(dst+i, _mm256_cvtph_ps(_mm256_cvtps_ph(_mm256_loadu_ps(src+i), _MM_FROUND_TO_NEAREST_INT|_MM_FROUND_NO_EXC)));
	for (int64_t i=n&~(int64_t)7; i<n; ++i)
		dst[i] = _cvtsh_ss(_cvtss_sh(src[i], _MM_FROUND_TO_NEAREST_INT|_MM_FROUND_NO_EXC));
}

/* ------------------------------------------------------------------ SGEMM
 * C[M][N] = A[M][K] . B[N][K]^T, fp32.  Every element is accumulated by FMAs in k order inside K chunks of KC, the chunks are added to C in order: the result does
 * not depend on the tile sizes, the thread count or the vector width -- the AVX2 (6 x 16) and AVX-512 (12 x 32, chosen at run time where the CPU has it)
 * micro-kernels are bit-identical (tests/test_oracle_ops.py).
 * Round 5 (the CPU baseline ran this at 183 GFLOP/s on 64 threads): per K chunk, A and B are packed ONCE by all threads into shared panel
 * buffers ([k][MR] / [k][NR], 8 x 8 in-register transposes for the K-contiguous sources), then all threads walk the (row block x column block) tiles of C; two
 * barriers per chunk instead of one per (column block, chunk), no per-tile re-packing.
 */
#define KC 384
#define MR2 6
#define NR2 16
#define MR5 12
#define NR5 32

static int g_isa = -1;      /* -1: choose at first use; 2 = AVX2 micro-kernel, 5 = AVX-512 */
void orc_set_isa(int isa) { g_isa = (isa == 5 && __builtin_cpu_supports("avx512f")) ? 5 : (isa == 2 ? 2 : -1); }
int orc_get_isa(void)
{
	if (g_isa < 0) g_isa = __builtin_cpu_supports("avx512f") ? 5 : 2;
	return g_isa;
}

/* rows [r0, r0 + w) of S ([rows][ld], K-contiguous), columns [0, kc) -> panel [k][R] (rows beyond w: zeros) */
static void pack_panel(int64_t kc, int w, int R, const float* S, int64_t ld, float* P)
{
	int64_t k = 0;
	for (; k + 8 <= kc; k += 8) {
		for (int r0 = 0; r0 < R; r0 += 8) {
			__m256 v[8];
			for (int i=0;i<8;++i) v[i] = r0 + i < w ? _mm256_loadu_ps(S + (int64_t)(r0 + i)*ld + k) : _mm256_setzero_ps();
			__m256 t0 = _mm256_unpacklo_ps(v[0], v[1]), t1 = _mm256_unpackhi_ps(v[0], v[1]), t2 = _mm256_unpacklo_ps(v[2], v[3]), t3 = _mm256_unpackhi_ps(v[2], v[3]);
			__m256 t4 = _mm256_unpacklo_ps(v[4], v[5]), t5 = _mm256_unpackhi_ps(v[4], v[5]), t6 = _mm256_unpacklo_ps(v[6], v[7]), t7 = _mm256_unpackhi_ps(v[6], v[7]);
			__m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE), u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
			__m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE), u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
			__m256 o[8] = { _mm256_permute2f128_ps(u0, u4, 0x20), _mm256_permute2f128_ps(u1, u5, 0x20), _mm256_permute2f128_ps(u2, u6, 0x20), _mm256_permute2f128_ps(u3, u7, 0x20),
			                _mm256_permute2f128_ps(u0, u4, 0x31), _mm256_permute2f128_ps(u1, u5, 0x31), _mm256_permute2f128_ps(u2, u6, 0x31), _mm256_permute2f128_ps(u3, u7, 0x31) };
			const int n = R - r0 < 8 ? R - r0 : 8;
			if (n == 8) for (int kk=0;kk<8;++kk) _mm256_storeu_ps(P + (k + kk)*R + r0, o[kk]);
			else { float tmp[8]; for (int kk=0;kk<8;++kk) { _mm256_storeu_ps(tmp, o[kk]); for (int i=0;i<n;++i) P[(k + kk)*R + r0 + i] = tmp[i]; } }
		}
	}
	for (; k < kc; ++k) for (int r=0;r<R;++r) P[k*R + r] = r < w ? S[(int64_t)r*ld + k] : 0.f;
}

static inline void ukernel_6x16(int64_t kc, const float* Ap, const float* Bp, float* C, int64_t ldc, int mr, int nr, int accumulate)
{
	__m256 c[MR2][2];
	for (int i=0;i<MR2;++i) { c[i][0]=_mm256_setzero_ps(); c[i][1]=_mm256_setzero_ps(); }
	for (int64_t k=0; k<kc; ++k) {
		__m256 b0 = _mm256_loadu_ps(Bp + k*NR2), b1 = _mm256_loadu_ps(Bp + k*NR2 + 8);
		for (int i=0;i<MR2;++i) {
			__m256 a = _mm256_broadcast_ss(Ap + k*MR2 + i);
			c[i][0] = _mm256_fmadd_ps(a, b0, c[i][0]);
			c[i][1] = _mm256_fmadd_ps(a, b1, c[i][1]);
		}
	}
	if (mr == MR2 && nr == NR2) {
		for (int i=0;i<MR2;++i) {
			float *cp = C + i*ldc;
			if (accumulate) {
				_mm256_storeu_ps(cp,   _mm256_add_ps(_mm256_loadu_ps(cp),   c[i][0]));
				_mm256_storeu_ps(cp+8, _mm256_add_ps(_mm256_loadu_ps(cp+8), c[i][1]));
			} else {
				_mm256_storeu_ps(cp, c[i][0]); _mm256_storeu_ps(cp+8, c[i][1]);
			}
		}
	} else {
		float tmp[MR2][NR2];
		for (int i=0;i<MR2;++i) { _mm256_storeu_ps(tmp[i], c[i][0]); _mm256_storeu_ps(tmp[i]+8, c[i][1]); }
		for (int i=0;i<mr;++i) for (int j=0;j<nr;++j) {
			if (accumulate) C[i*ldc+j] += tmp[i][j]; else C[i*ldc+j] = tmp[i][j];
		}
	}
}

__attribute__((target("avx512f")))
static void ukernel_12x32(int64_t kc, const float* Ap, const float* Bp, float* C, int64_t ldc, int mr, int nr, int accumulate)
{
	__m512 c[MR5][2];
	for (int i=0;i<MR5;++i) { c[i][0]=_mm512_setzero_ps(); c[i][1]=_mm512_setzero_ps(); }
	for (int64_t k=0; k<kc; ++k) {
		const __m512 b0 = _mm512_loadu_ps(Bp + k*NR5), b1 = _mm512_loadu_ps(Bp + k*NR5 + 16);
		for (int i=0;i<MR5;++i) {
			const __m512 a = _mm512_set1_ps(Ap[k*MR5 + i]);
			c[i][0] = _mm512_fmadd_ps(a, b0, c[i][0]);
			c[i][1] = _mm512_fmadd_ps(a, b1, c[i][1]);
		}
	}
	const __mmask16 m0 = nr >= 16 ? 0xFFFF : (__mmask16)((1u << nr) - 1), m1 = nr >= 32 ? 0xFFFF : (nr > 16 ? (__mmask16)((1u << (nr - 16)) - 1) : 0);
	for (int i=0;i<mr;++i) {
		float *cp = C + i*ldc;
		if (accumulate) {
			_mm512_mask_storeu_ps(cp,      m0, _mm512_add_ps(_mm512_maskz_loadu_ps(m0, cp),      c[i][0]));
			_mm512_mask_storeu_ps(cp + 16, m1, _mm512_add_ps(_mm512_maskz_loadu_ps(m1, cp + 16), c[i][1]));
		} else {
			_mm512_mask_storeu_ps(cp, m0, c[i][0]); _mm512_mask_storeu_ps(cp + 16, m1, c[i][1]);
		}
	}
}

/* B operand given by a callback (implicit im2col, o_ops.c): fill panel P[k][R] with columns j0 .. j0 + w - 1 (zero-padded to R) for k in [k0, k0 + kc) */
typedef void (*orc_bpack_fn)(void* ctx, int64_t j0, int w, int64_t k0, int64_t kc, int R, float* P);

static void sgemm_core(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, orc_bpack_fn bpack, void* bctx, float* C, int64_t ldc)
{
	if (M<=0 || N<=0) return;
	if (K<=0) { for (int64_t i=0;i<M;++i) memset(C+i*ldc, 0, (size_t)N*sizeof(float)); return; }
	const int nth = orc_get_threads();
	const int isa = orc_get_isa();
	const int MR = isa == 5 ? MR5 : MR2, NR = isa == 5 ? NR5 : NR2;
	/* columns in chunks of NC (the packed B chunk, NC x KC floats, stays near the cores; a convolution's N is its pixel count: 10^6 at 1024 x 1024) */
	const int64_t NC = nth > 64 ? 16384 : 8192;
	const int64_t npm = (M + MR - 1) / MR, ncmax = N < NC ? N : NC, npn_max = (ncmax + NR - 1) / NR;
	float *Ap = (float*)orc_balloc((size_t)npm*MR*KC*sizeof(float)), *Bp = (float*)orc_balloc((size_t)npn_max*NR*KC*sizeof(float));
	if (!Ap || !Bp) { orc_bfree(Ap); orc_bfree(Bp); return; }
	/* tiles of C inside a chunk: TM row panels x TN column panels, sized so that a tile's A panels (TM MR KC floats) stay in L2 and there are >= 4 tiles per thread */
	int64_t TM = 96 / MR, TN = 512 / NR;
	while (TN > 2 && ((npm + TM - 1) / TM) * ((npn_max + TN - 1) / TN) < 4 * (int64_t)nth) TN /= 2;
	while (TM > 2 && ((npm + TM - 1) / TM) * ((npn_max + TN - 1) / TN) < 4 * (int64_t)nth) TM /= 2;
	const int64_t ntm = (npm + TM - 1) / TM;
	#pragma omp parallel num_threads(nth)
	for (int64_t pc=0; pc<K; pc+=KC) {
		const int64_t kc = K-pc < KC ? K-pc : KC;
		#pragma omp for schedule(static)
		for (int64_t ip=0; ip<npm; ++ip) {
			const int64_t i = ip*MR; const int w = (int)(M-i < MR ? M-i : MR);
			pack_panel(kc, w, MR, A + i*lda + pc, lda, Ap + ip*kc*MR);
		}
		for (int64_t jc=0; jc<N; jc+=NC) {
			const int64_t nc = N-jc < NC ? N-jc : NC, npn = (nc + NR - 1) / NR, ntn = (npn + TN - 1) / TN;
			#pragma omp for schedule(static)
			for (int64_t jp=0; jp<npn; ++jp) {
				const int64_t j = jc + jp*NR; const int w = (int)(N-j < NR ? N-j : NR);
				if (bpack) bpack(bctx, j, w, pc, kc, NR, Bp + jp*kc*NR);
				else pack_panel(kc, w, NR, B + j*ldb + pc, ldb, Bp + jp*kc*NR);
			}
			/* (implicit barrier: every panel of this chunk is packed) */
			#pragma omp for schedule(dynamic,1) collapse(2)
			for (int64_t tm=0; tm<ntm; ++tm)
			for (int64_t tn=0; tn<ntn; ++tn) {
				const int64_t ip1 = (tm+1)*TM < npm ? (tm+1)*TM : npm, jp1 = (tn+1)*TN < npn ? (tn+1)*TN : npn;
				for (int64_t jp=tn*TN; jp<jp1; ++jp) {
					const int64_t j = jc + jp*NR;
					const int nr = (int)(N - j < NR ? N - j : NR);
					for (int64_t ip=tm*TM; ip<ip1; ++ip) {
						const int mr = (int)(M - ip*MR < MR ? M - ip*MR : MR);
						if (isa == 5) ukernel_12x32(kc, Ap + ip*kc*MR, Bp + jp*kc*NR, C + ip*MR*ldc + j, ldc, mr, nr, pc>0);
						else ukernel_6x16(kc, Ap + ip*kc*MR, Bp + jp*kc*NR, C + ip*MR*ldc + j, ldc, mr, nr, pc>0);
					}
				}
			}
			/* (implicit barrier: the B chunk buffer is free again) */
		}
	}
	orc_bfree(Ap); orc_bfree(Bp);
}

void orc_sgemm_nt(int64_t M, int64_t N, int64_t K,
	const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc)
{
	sgemm_core(M, N, K, A, lda, B, ldb, NULL, NULL, C, ldc);
}

/* C[M][N] = A[M][K] . Bgen[N][K]^T with the rows of Bgen produced panel by panel by `bpack` (never materialised) */
void orc_sgemm_nt_gen(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, orc_bpack_fn bpack, void* bctx, float* C, int64_t ldc)
{
	sgemm_core(M, N, K, A, lda, NULL, 0, bpack, bctx, C, ldc);
}

/* ------------------------------------------------------------------ synthetic weights
 * Pure-integer Irwin-Hall(4) "normal" from a splitmix64 stream keyed by (seed, name, i):
 * bit-identical on CPU and GPU (one fp32 multiply + one fp32 add, no transcendental).
 * i is the element index in the REFERENCE layout (ne[0] fastest).
 *   value_i = offset + (float)(s_i - 131070) * kf,   kf = (float)(scale*sqrt(3)/65536)
 * Rule per parameter (orc_synth_rule):
 *   1-D "*.weight" (norm gamma)  : offset 1, scale 0.1
 *   1-D "*.bias"                 : offset 0, scale 0.05
 *   >=2-D weights                : offset 0, scale 1/sqrt(fan_in), fan_in = ne0 (linear,
 *                                  embedding, text_proj) or ne0*ne1*ne2 (conv)
 */
static inline uint64_t mix64(uint64_t z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
	z ^= z >> 27; z *= 0x94D049BB133111EBULL;
	z ^= z >> 31; return z;
}

static uint64_t fnv1a64(const char* s)
{
	uint64_t h = 0xCBF29CE484222325ULL;
	for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001B3ULL; }
	return h;
}

void orc_synth_fill(float* out, int64_t n, uint64_t seed, const char* name,
	float offset, float scale, int round_f16)
{
	const uint64_t key = mix64(fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15ULL));
	const float kf = (float)((double)scale * 1.7320508075688772 / 65536.0);
	#pragma omp parallel for schedule(static) if (n > 65536)
	for (int64_t i=0; i<n; ++i) {
		uint64_t u = mix64(key + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
		int s = (int)(u & 0xFFFF) + (int)((u>>16)&0xFFFF) + (int)((u>>32)&0xFFFF) + (int)(u>>48);
		float v = (float)(s - 131070) * kf;
		out[i] = offset + v;
	}
	if (round_f16) orc_round_f16(out, n);
}

static int str_ends(const char* s, const char* suf)
{
	size_t ls=strlen(s), lf=strlen(suf);
	return ls>=lf && !strcmp(s+ls-lf, suf);
}

void orc_synth_rule(const char* name, int type, const int64_t ne[4], float* offset, float* scale)
{
	int nd = 4; while (nd>1 && ne[nd-1]==1) nd--;
	(void)type;
	if (nd == 1) {
		if (str_ends(name, ".bias")) { *offset=0; *scale=0.05f; }
		else { *offset=1; *scale=0.1f; }
	} else {
		double fan_in = (nd >= 3) ? (double)ne[0]*ne[1]*ne[2] : (double)ne[0];
		*offset = 0; *scale = (float)(1.0/sqrt(fan_in));
	}
}

struct OParams {
	uint64_t seed;
	int n, cap;
	OParam **v;   /* entries are individually allocated: handed-out pointers stay valid */
};

OParams* orc_params_new(uint64_t seed)
{
	OParams *P = (OParams*)calloc(1, sizeof(OParams));
	P->seed = seed;
	return P;
}

void orc_params_free(OParams* P)
{
	if (!P) return;
	for (int i=0;i<P->n;++i) { free(P->v[i]->name); free(P->v[i]->d); free(P->v[i]); }
	free(P->v); free(P);
}

static OParam* params_find(OParams* P, const char* name)
{
	for (int i=P->n-1; i>=0; --i) if (!strcmp(P->v[i]->name, name)) return P->v[i];
	return NULL;
}

static OParam* params_add(OParams* P, const char* name, int type, const int64_t ne[4])
{
	if (P->n == P->cap) { P->cap = P->cap ? P->cap*2 : 256; P->v = (OParam**)realloc(P->v, sizeof(OParam*)*P->cap); }
	OParam *e = (OParam*)calloc(1, sizeof(OParam));
	P->v[P->n++] = e;
	e->name = strdup(name); e->type = type;
	memcpy(e->ne, ne, sizeof(e->ne));
	int64_t n = ne[0]*ne[1]*ne[2]*ne[3];
	if (posix_memalign((void**)&e->d, 64, (size_t)(n>0?n:1)*sizeof(float))) e->d = NULL;
	return e;
}

int orc_params_set(OParams* P, const char* name, int type,
	int64_t n0, int64_t n1, int64_t n2, int64_t n3, const float* data)
{
	int64_t ne[4] = {n0,n1,n2,n3};
	OParam *e = params_find(P, name);
	if (e) {
		if (e->ne[0]*e->ne[1]*e->ne[2]*e->ne[3] != n0*n1*n2*n3) return -1;
		memcpy(e->ne, ne, sizeof(ne)); e->type = type;
	} else e = params_add(P, name, type, ne);
	int64_t n = n0*n1*n2*n3;
	memcpy(e->d, data, (size_t)n*sizeof(float));
	if (type == ORC_F16) orc_round_f16(e->d, n);
	return 1;
}

const OParam* orc_params_get(OParams* P, const char* name, int type,
	int64_t n0, int64_t n1, int64_t n2, int64_t n3)
{
	int64_t ne[4] = {n0,n1,n2,n3};
	OParam *e = params_find(P, name);
	int64_t n = n0*n1*n2*n3;
	if (e) {
		/* like tstore_tensor_read (src/mlblock.c:243): element count is what is checked */
		if (e->ne[0]*e->ne[1]*e->ne[2]*e->ne[3] != n) {
			fprintf(stderr, "oracle: param '%s' size mismatch\n", name);
			return NULL;
		}
		return e;
	}
	e = params_add(P, name, type, ne);
	float off, sc;
	orc_synth_rule(name, type, ne, &off, &sc);
	orc_synth_fill(e->d, n, P->seed, name, off, sc, type == ORC_F16);
	return e;
}

int orc_params_count(const OParams* P) { return P->n; }
const OParam* orc_params_at(const OParams* P, int i) { return (i>=0 && i<P->n) ? P->v[i] : NULL; }
