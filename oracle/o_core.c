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
OT* ot_new(int64_t n0, int64_t n1, int64_t n2, int64_t n3)
{
	OT *t = (OT*)calloc(1, sizeof(OT));
	t->ne[0]=n0; t->ne[1]=n1; t->ne[2]=n2; t->ne[3]=n3;
	int64_t n = n0*n1*n2*n3;
	if (posix_memalign((void**)&t->d, 64, (size_t)(n>0?n:1)*sizeof(float))) { free(t); return NULL; }
	return t;
}

OT* ot_from(const float* src, int64_t n0, int64_t n1, int64_t n2, int64_t n3)
{
	OT *t = ot_new(n0,n1,n2,n3);
	memcpy(t->d, src, (size_t)ot_nel(t)*sizeof(float));
	return t;
}

void ot_free(OT* t) { if (t) { free(t->d); free(t); } }

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

/* ------------------------------------------------------------------ SGEMM
 * C[M][N] = A[M][K] . B[N][K]^T, fp32, k-ordered accumulation inside KC blocks.
 * BLIS-style: pack B panels (NR=16 wide), pack A panels (MR=6 tall), 6x16 AVX2 micro-kernel.
 */
#define MR 6
#define NR 16
#define KC 384
#define MC 120   /* multiple of MR */
#define NC 2048  /* multiple of NR */

static void pack_B(int64_t kc, int64_t nc, const float* B, int64_t ldb, float* Bp)
{	/* B is [N][K]; panel j holds kc x NR with layout [k][NR] */
	for (int64_t j=0; j<nc; j+=NR) {
		int64_t nr = nc-j < NR ? nc-j : NR;
		for (int64_t k=0; k<kc; ++k) {
			for (int64_t jj=0; jj<nr; ++jj) Bp[k*NR+jj] = B[(j+jj)*ldb + k];
			for (int64_t jj=nr; jj<NR; ++jj) Bp[k*NR+jj] = 0;
		}
		Bp += kc*NR;
	}
}

static void pack_A(int64_t kc, int64_t mc, const float* A, int64_t lda, float* Ap)
{	/* A is [M][K]; panel i holds kc x MR with layout [k][MR] */
	for (int64_t i=0; i<mc; i+=MR) {
		int64_t mr = mc-i < MR ? mc-i : MR;
		for (int64_t k=0; k<kc; ++k) {
			for (int64_t ii=0; ii<mr; ++ii) Ap[k*MR+ii] = A[(i+ii)*lda + k];
			for (int64_t ii=mr; ii<MR; ++ii) Ap[k*MR+ii] = 0;
		}
		Ap += kc*MR;
	}
}

static inline void ukernel_6x16(int64_t kc, const float* Ap, const float* Bp,
	float* C, int64_t ldc, int mr, int nr, int accumulate)
{
	__m256 c[MR][2];
	for (int i=0;i<MR;++i) { c[i][0]=_mm256_setzero_ps(); c[i][1]=_mm256_setzero_ps(); }
	for (int64_t k=0; k<kc; ++k) {
		__m256 b0 = _mm256_loadu_ps(Bp + k*NR), b1 = _mm256_loadu_ps(Bp + k*NR + 8);
		for (int i=0;i<MR;++i) {
			__m256 a = _mm256_broadcast_ss(Ap + k*MR + i);
			c[i][0] = _mm256_fmadd_ps(a, b0, c[i][0]);
			c[i][1] = _mm256_fmadd_ps(a, b1, c[i][1]);
		}
	}
	if (mr == MR && nr == NR) {
		for (int i=0;i<MR;++i) {
			float *cp = C + i*ldc;
			if (accumulate) {
				_mm256_storeu_ps(cp,   _mm256_add_ps(_mm256_loadu_ps(cp),   c[i][0]));
				_mm256_storeu_ps(cp+8, _mm256_add_ps(_mm256_loadu_ps(cp+8), c[i][1]));
			} else {
				_mm256_storeu_ps(cp, c[i][0]); _mm256_storeu_ps(cp+8, c[i][1]);
			}
		}
	} else {
		float tmp[MR][NR];
		for (int i=0;i<MR;++i) { _mm256_storeu_ps(tmp[i], c[i][0]); _mm256_storeu_ps(tmp[i]+8, c[i][1]); }
		for (int i=0;i<mr;++i) for (int j=0;j<nr;++j) {
			if (accumulate) C[i*ldc+j] += tmp[i][j]; else C[i*ldc+j] = tmp[i][j];
		}
	}
}

void orc_sgemm_nt(int64_t M, int64_t N, int64_t K,
	const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc)
{
	if (M<=0 || N<=0) return;
	if (K<=0) { for (int64_t i=0;i<M;++i) memset(C+i*ldc, 0, (size_t)N*sizeof(float)); return; }
	int nth = orc_get_threads();
	float *Bp = NULL;
	if (posix_memalign((void**)&Bp, 64, (size_t)KC*NC*sizeof(float))) return;
	for (int64_t jc=0; jc<N; jc+=NC) {
		int64_t nc = N-jc < NC ? N-jc : NC;
		for (int64_t pc=0; pc<K; pc+=KC) {
			int64_t kc = K-pc < KC ? K-pc : KC;
			/* pack B (parallel over panels) */
			int64_t npan = (nc+NR-1)/NR;
			#pragma omp parallel for schedule(static) num_threads(nth)
			for (int64_t jp=0; jp<npan; ++jp) {
				int64_t j = jp*NR, w = nc-j < NR ? nc-j : NR;
				pack_B(kc, w, B + (jc+j)*ldb + pc, ldb, Bp + jp*kc*NR);
			}
			/* 2-D tile parallelism (row blocks x column chunks): convs have few output channels (M) but many
			 * pixels (N), linears the opposite; each tile packs its own A block (<1 % overhead) */
			const int64_t nblk = (M+MC-1)/MC, JB = 256, njb = (nc+JB-1)/JB;
			#pragma omp parallel num_threads(nth)
			{
				float *Ap = NULL;
				if (posix_memalign((void**)&Ap, 64, (size_t)KC*MC*sizeof(float))) Ap = NULL;
				#pragma omp for schedule(dynamic,1) collapse(2)
				for (int64_t ib=0; ib<nblk; ++ib)
				for (int64_t jb=0; jb<njb; ++jb) {
					int64_t ic = ib*MC, mc = M-ic < MC ? M-ic : MC;
					int64_t j0 = jb*JB, j1 = j0+JB < nc ? j0+JB : nc;
					pack_A(kc, mc, A + ic*lda + pc, lda, Ap);
					for (int64_t jr=j0; jr<j1; jr+=NR) {
						int nr = (int)(nc-jr < NR ? nc-jr : NR);
						for (int64_t ir=0; ir<mc; ir+=MR) {
							int mr = (int)(mc-ir < MR ? mc-ir : MR);
							ukernel_6x16(kc, Ap + (ir/MR)*kc*MR, Bp + (jr/NR)*kc*NR,
								C + (ic+ir)*ldc + jc+jr, ldc, mr, nr, pc>0);
						}
					}
				}
				free(Ap);
			}
		}
	}
	free(Bp);
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
