/* Host-side graph builder + plan executor (C), the MI355X re-creation of the reference's
 * MLCtx runtime (src/mlblock.h:44-160, src/mlblock.c:54-345).
 *
 *   reference                         here
 *   ---------                         ----
 *   two ggml contexts (cp, cc)        MLParam table (device-resident weights) + MLTensor list
 *   ggml_cgraph + gallocr             recorded launch plan (MLOp[]) + arena with exact-size reuse
 *   mlctx_load_prep name resolution   same backward walk over (name, BLOCK_BEGIN) records
 *   mlctx_tstore_load + tensor_set    mlctx_param_set (repack to device layout) / mlctx_params_synth
 *   ggml_backend_graph_compute        mlctx_compute: replay the plan on a HIP stream, or one
 *                                     hipGraphLaunch when MLB_F_HIPGRAPH is set
 */
#include "mlblock_int.h"
#include <math.h>
#include <stdarg.h>
#include <time.h>

#define CHUNK_BYTES   ((size_t)64 << 20)
#define BIG_BYTES     ((size_t)8 << 20)
#define ALIGN_UP(x,a) (((x) + (a) - 1) / (a) * (a))

#define VARIANT_STREAMK 20      /* tile_variant (1-based) of the stream-K ping-pong tile (256 x 256) */
#define VARIANT_STREAMK2 29     /* the same on the 128 x 320 tile (round 4) */
#define IS_STREAMK(v) ((v) == VARIANT_STREAMK || (v) == VARIANT_STREAMK2)
#define VARIANT_TT 31           /* two tiles in flight per CU (gemm_tt.hip, round 5) */
#define VARIANT_SKINNY 30       /* the skinny-M weight-streaming kernel (gemm_skinny.hpp): always runs through the split-K workspace */
#define SK_FLAG_WORDS 4096      /* stream-K: a flag per persistent block (256); split-K reduced in the launch: a ticket counter per output tile; word 4095: sticky give-up */
#define LN_CNT_WORDS 8192       /* LayerNorm fold: word 8191 = sticky give-up (round 6: self-tagged records instead of arrival / departure counters; a tile takes its tag from its own record of the op's previous launch: nothing else lives here) */

#define VEC_PUSH(C, arr, n, cap, T) \
	(((n) == (cap) ? ((cap) = (cap) ? (cap)*2 : 64, (arr) = (T*)realloc((arr), sizeof(T)*(cap))) : 0), &(arr)[(n)++])

static double now_s(void)
{
	struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec*1e-9;
}

int mlctx_fail(MLCtx* C, const char* fmt, ...)
{
	char buf[400];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	if (C && !C->err) C->err = -1;
	return mlsd_set_error(-1, "%s%s%s", C ? C->name : "", C ? ": " : "", buf);
}

/* ------------------------------------------------------------------ lifecycle */
MLB_API MLCtx* mlctx_new(void* stream)
{
	MLCtx *C = (MLCtx*)calloc(1, sizeof(MLCtx));
	C->stream = stream;
	C->wtype = MLT_F16;
	C->dry = mlsd_runtime_is_dry();
	return C;
}

static void ctx_reset_(MLCtx* C);
static void wstream_free(MLCtx* C);
static void ctx_reset(MLCtx* C)
{	/* memory handed out by the dry runtime is host memory: release it the same way even if the mode changed since */
	const int was = mlsd_runtime_is_dry();
	if (C->dry != was) mlsd_runtime_dry(C->dry);
	ctx_reset_(C);
	if (C->dry != was) mlsd_runtime_dry(was);
	C->dry = was;           /* what is built from here on follows the current mode */
}

static void ctx_reset_(MLCtx* C)
{
	if (C->graph_exec) { mlsd_graph_destroy(C->graph_exec); C->graph_exec = NULL; }
	for (int i=0;i<C->n_tensors;++i) free(C->tensors[i]);
	C->n_tensors = 0; C->n_inputs = 0; C->result = NULL;
	for (int i=0;i<C->n_names;++i) free(C->names[i].name);
	C->n_names = 0;
	for (int i=0;i<C->n_params;++i) free(C->params[i].key);
	C->n_params = 0;
	C->n_ops = 0;
	for (int i=0;i<C->n_chunks;++i) mlsd_free(C->chunks[i]);
	if (C->splitk_ws) { mlsd_free(C->splitk_ws); C->splitk_ws = NULL; C->splitk_ws_bytes = 0; }
	if (C->sk_flags) { mlsd_free(C->sk_flags); C->sk_flags = NULL; }
	if (C->ln_cnt) { mlsd_free(C->ln_cnt); C->ln_cnt = NULL; }
	if (C->ln_ws) { mlsd_free(C->ln_ws); C->ln_ws = NULL; C->ln_ws_bytes = 0; }
	wstream_free(C);
	C->n_chunks = 0; C->cur = NULL; C->cur_left = 0; C->n_free = 0;
	C->mem_compute = C->mem_params = C->mem_live = C->mem_peak_live = 0;
	C->err = 0; C->prepared = 0; C->tuned = 0; C->n_tune_miss = 0; C->n_tune_near = 0; C->static_valid = 0; C->n_once = 0;
	C->n_ln_fused = 0; C->n_ln_alias = 0; C->n_gn_fused = 0;
	memset(&C->kvb, 0, sizeof(C->kvb));
	memset(&C->epb, 0, sizeof(C->epb));
	memset(&C->info, 0, sizeof(C->info));
}

MLB_API void mlctx_destroy(MLCtx* C)
{
	if (!C) return;
	mlsd_stream_sync(C->stream);
	ctx_reset(C);
	free(C->ops); free(C->tensors); free(C->inputs); free(C->names); free(C->params); free(C->chunks); free(C->freel);
	free(C);
}

MLB_API void mlctx_free(MLCtx* C) { mlctx_destroy(C); }
MLB_API MLTensor* mlctx_split_add(MLCtx* C, MLTensor* t) { (void)C; return t; }

MLB_API void mlctx_begin(MLCtx* C, const char* name)
{
	mlsd_stream_sync(C->stream);
	ctx_reset(C);
	snprintf(C->name, sizeof(C->name), "%s", name ? name : "");
}

MLB_API void mlctx_end(MLCtx* C) { mlsd_stream_sync(C->stream); ctx_reset(C); }
MLB_API void mlctx_set_tprefix(MLCtx* C, const char* p) { snprintf(C->tprefix, sizeof(C->tprefix), "%s", p ? p : ""); }
MLB_API void mlctx_set_flags(MLCtx* C, int f) { C->flags = f; }
MLB_API void mlctx_set_wtype(MLCtx* C, int t) { C->wtype = t; }
MLB_API MLTensor* mlctx_result(MLCtx* C) { return C->result; }
MLB_API int mlctx_sync(MLCtx* C) { return mlsd_stream_sync(C->stream) ? -1 : 1; }

MLB_API int mlctx_ln_fused(const MLCtx* C) { return C ? C->n_ln_fused : 0; }
MLB_API int mlctx_ln_alias_refused(const MLCtx* C) { return C ? C->n_ln_alias : 0; }   /* LayerNorm folds refused because the fp16 rows would have overwritten an operand of their producer */
/* CUs the plan's stream may use (a CU-masked stream, a partitioned device); 0 = the whole device.  Below 256 the plan is built without in-launch hand-offs
 * (stream-K, LayerNorm fold): their blocks wait for partners that a masked stream may never make resident.  Call before mlctx_prep. */
MLB_API void mlctx_set_cus(MLCtx* C, int n) { if (C) C->cu_budget = n; }   /* LayerNorms of the plan that run at the end of their producers' launches */

/* Stream-K launches hand partial tiles over inside the launch, LayerNorm-ending launches exchange row statistics; a block that gives up waiting (bounded spin: a
 * partner never became resident, e.g. the CUs are shared with another process or the stream is CU-masked) raises a sticky word beside the flags.  Called by the drivers
 * where they read their results back: 0 = clean, < 0 = at least one hand-off of this plan timed out since the last check.
 * After a give-up the per-block flags / arrival and departure counters are whatever the timed-out launch left (a late contributor sets a flag its owner no longer
 * clears; a late tile leaves (arrivals, departures) = (1, 1)): the NEXT launch would consume a stale slab or stale statistics without raising the word again.
 * So on a give-up the stream is drained and BOTH blocks are zeroed entirely before the error is returned. */
MLB_API int mlctx_handoff_check(MLCtx* C)
{
	if (!C) return 0;
	unsigned w = 0, w2 = 0;
	if (C->sk_flags && (mlsd_memcpy(&w, C->sk_flags + 4095, 4, 1, C->stream) || mlsd_stream_sync(C->stream))) return -1;
	if (C->ln_cnt && (mlsd_memcpy(&w2, C->ln_cnt + 8191, 4, 1, C->stream) || mlsd_stream_sync(C->stream))) return -1;
	if (!w && !w2) return 0;
	mlsd_stream_sync(C->stream);
	if (C->sk_flags) mlsd_memset(C->sk_flags, 0, SK_FLAG_WORDS * 4, C->stream);
	if (C->ln_cnt) mlsd_memset(C->ln_cnt, 0, LN_CNT_WORDS * 4, C->stream);      /* the sticky word ... */
	if (C->ln_ws) mlsd_memset(C->ln_ws, 0, C->ln_ws_bytes, C->stream);              /* ... and every record with them: a half-written generation must not meet its own tag again */
	mlsd_stream_sync(C->stream);
	return mlsd_set_error(-8, "an in-launch hand-off (%s) timed out (block not resident: is the GPU shared with another process?); results of this plan are invalid",
	                      w ? "stream-K" : "LayerNorm statistics");
}

/* number of ops of the plan that hand data over inside a launch (stream-K tiles, LayerNorms ended in their producers): 0 = nothing to check */
MLB_API int mlctx_handoff_ops(const MLCtx* C)
{
	int n = 0;
	if (!C || !C->prepared) return 0;
	for (int i=0;i<C->n_ops;++i) {
		const MLOp *o = &C->ops[i];
		if (o->kind != OP_GEMM) continue;
		if ((o->u.gemm.ln_y16 && mlsd_gemm_ln_fused(&o->u.gemm) == 1) || (IS_STREAMK(o->u.gemm.tile_variant) && o->u.gemm.sk_flags) || mlsd_gemm_splitk_parallel(&o->u.gemm)) ++n;
	}
	return n;
}

/* The plan without its in-launch hand-offs: stream-K launches lose their flag words, so the launcher runs them on the plain tile of the same shape
 * (mlsd_gemm: sk_eligible fails); LayerNorms that were ended in their producers get their own launch back (the op and its output buffer were kept).  A captured
 * hipGraph is dropped (the next compute captures the new launches).  Results change in the last bits (other fp32 summation order in the former stream-K tiles).
 * Returns the number of ops changed.  One-way: a context that timed out once stays on the hand-off-free plan. */
static int g_handoff_retries = 0;
MLB_API int mlctx_handoff_retries(void) { return g_handoff_retries; }

MLB_API int mlctx_handoffs_off(MLCtx* C)
{
	int n = 0;
	if (!C || !C->prepared) return 0;
	mlsd_stream_sync(C->stream);
	for (int i=0;i<C->n_ops;++i) {
		MLOp *o = &C->ops[i];
		if (o->kind != OP_GEMM) continue;
		mlsd_gemm_args *g = &o->u.gemm;
		if (g->ln_y16 && mlsd_gemm_ln_fused(g) == 1) {      /* (a LayerNorm in a split-K reduce pass waits for nobody: it stays) */
			g->ln_y16 = NULL; g->ln_gamma = g->ln_beta = NULL; g->ln_ws = NULL; g->ln_cnt = NULL;
			if (i + 1 < C->n_ops && C->ops[i+1].kind == OP_LN && C->ops[i+1].fused) { C->ops[i+1].fused = 0; C->n_ln_fused--; }
			++n;
			if (o->folded_from) { g->tile_variant = o->folded_from < 0 ? 0 : o->folded_from; o->folded_from = 0; }      /* moved to the 128 x 160 kernel only for this LayerNorm: back to its own tile */
		}
		if ((IS_STREAMK(g->tile_variant) || g->ksplit > 1) && g->sk_flags) { g->sk_flags = NULL; ++n; }     /* (split-K: back to the two-launch form) */
	}
	if (n && C->graph_exec) { mlsd_graph_destroy(C->graph_exec); C->graph_exec = NULL; }
	C->static_valid = 0;      /* the timed-out launch may have been a hoisted step-invariant op (stream-K is not excluded for `once` ops): every caller's re-run recomputes them */
	return n;
}

/* mlctx_compute for callers that read the result back right away (the builder-level run functions): compute, and if the plan has in-launch hand-offs, drain the
 * stream and check them; after a give-up the evaluation is run AGAIN in the same process on the hand-off-free plan (counted: mlctx_handoff_retries), and only a
 * failure of that second run is an error. */
MLB_API int mlctx_compute_checked(MLCtx* C)
{
	if (mlctx_compute(C) < 0) return -1;
	if (!mlctx_handoff_ops(C)) return 1;
	if (mlsd_stream_sync(C->stream)) return -1;
	if (mlctx_handoff_check(C) == 0) return 1;
	mlctx_handoffs_off(C);
	g_handoff_retries++;
	if (mlctx_compute(C) < 0 || mlsd_stream_sync(C->stream)) return -1;
	return mlctx_handoff_check(C) == 0 ? 1 : -1;
}

/* test hook: raise the sticky give-up word as a timed-out launch would (what = 0: stream-K flags, 1: LayerNorm counters) */
MLB_API int mlctx_debug_raise_giveup(MLCtx* C, int what)
{
	const unsigned v = 0xDEADu;
	if (what == 0 && C->sk_flags) return mlsd_memcpy(C->sk_flags + 4095, &v, 4, 0, C->stream) || mlsd_stream_sync(C->stream) ? -1 : 1;
	if (what == 1 && C->ln_cnt) return mlsd_memcpy(C->ln_cnt + 8191, &v, 4, 0, C->stream) || mlsd_stream_sync(C->stream) ? -1 : 1;
	return 0;
}

/* ------------------------------------------------------------------ device memory */
/* A released block may be handed out again only to a writer that runs AFTER the block's last reader.
 * Ops run in recorded order, so a block released when `rel_op` ops had been recorded is safe for any op
 * with index >= rel_op.  `writer_op` is the index of the op that will write the new allocation: the next
 * op to be recorded for ordinary allocations (INT32_MAX), or the already-recorded producer when a GEMM
 * output is bound late (mlt_need16/32). */
static void* dalloc_ex(MLCtx* C, size_t nbytes, int is_param, int writer_op)
{
	nbytes = ALIGN_UP(nbytes ? nbytes : 256, 256);
	if (!is_param) {
		for (int i=C->n_free-1; i>=0; --i) if (C->freel[i].size == nbytes && C->freel[i].rel_op <= writer_op) {
			void *p = C->freel[i].ptr;
			C->freel[i] = C->freel[--C->n_free];
			C->mem_live += nbytes; if (C->mem_live > C->mem_peak_live) C->mem_peak_live = C->mem_live;
			return p;
		}
	}
	void *p = NULL;
	if (nbytes >= BIG_BYTES) {
		if (mlsd_malloc(&p, nbytes)) { mlctx_fail(C, "device allocation of %zu bytes failed", nbytes); return NULL; }
		*VEC_PUSH(C, C->chunks, C->n_chunks, C->cap_chunks, void*) = p;
	} else {
		if (C->cur_left < nbytes) {
			void *c = NULL;
			if (mlsd_malloc(&c, CHUNK_BYTES)) { mlctx_fail(C, "device chunk allocation failed"); return NULL; }
			*VEC_PUSH(C, C->chunks, C->n_chunks, C->cap_chunks, void*) = c;
			C->cur = (char*)c; C->cur_left = CHUNK_BYTES;
		}
		p = C->cur; C->cur += nbytes; C->cur_left -= nbytes;
	}
	if (is_param == 1) C->mem_params += nbytes;
	else { C->mem_compute += nbytes; C->mem_live += nbytes; if (C->mem_live > C->mem_peak_live) C->mem_peak_live = C->mem_live; }
	return p;
}

void* mlctx_dalloc(MLCtx* C, size_t nbytes, int is_param) { return dalloc_ex(C, nbytes, is_param, 0x7fffffff); }

static int is_virtual(const MLCtx* C, const void* p) { return C->pstream && (const char*)p >= MLW_VBASE && (const char*)p < MLW_VBASE + C->pv_size; }

void* mlctx_walloc(MLCtx* C, size_t nbytes)
{
	if (!C->pstream) return dalloc_ex(C, nbytes, 1, 0x7fffffff);
	nbytes = ALIGN_UP(nbytes ? nbytes : 256, 256);
	MLWAlloc *a = VEC_PUSH(C, C->pv_allocs, C->n_pv, C->cap_pv, MLWAlloc);
	a->voff = C->pv_size; a->bytes = nbytes; a->moff = a->voff;
	C->pv_size += nbytes;
	return MLW_VBASE + a->voff;
}

void mlctx_drelease(MLCtx* C, void* p, size_t nbytes)
{
	if (!p || !nbytes) return;
	nbytes = ALIGN_UP(nbytes, 256);
	MLFreeBlk *b = VEC_PUSH(C, C->freel, C->n_free, C->cap_free, MLFreeBlk);
	b->ptr = p; b->size = nbytes; b->rel_op = C->n_ops;
	C->mem_live -= nbytes;
}

/* ------------------------------------------------------------------ tensors */
MLTensor* mlt_new(MLCtx* C, int n, int h, int w, int c)
{
	MLTensor *t = (MLTensor*)calloc(1, sizeof(MLTensor));
	t->n = n; t->h = h; t->w = w; t->c = c; t->prod = -1; t->def_op = C->n_ops;
	if (h > 1 || 0) { t->ne[0]=w; t->ne[1]=h; t->ne[2]=c; t->ne[3]=n; }
	else { t->ne[0]=c; t->ne[1]=w; t->ne[2]=n; t->ne[3]=1; }
	*VEC_PUSH(C, C->tensors, C->n_tensors, C->cap_tensors, MLTensor*) = t;
	return t;
}

MLB_API void mlctx_tensor_shape(const MLTensor* t, int64_t ne[4]) { memcpy(ne, t->ne, sizeof(t->ne)); }

MLOp* mlctx_op_new(MLCtx* C, MLOpKind kind, const char* label)
{
	MLOp *op = VEC_PUSH(C, C->ops, C->n_ops, C->cap_ops, MLOp);
	memset(op, 0, sizeof(*op));
	op->kind = kind;
	snprintf(op->label, sizeof(op->label), "%s", label ? label : "");
	return op;
}

float* mlt_need32(MLCtx* C, MLTensor* t)
{
	if (!t) return NULL;
	if (t->d32) return t->d32;
	if (t->released) { mlctx_fail(C, "use of released tensor %s", t->name); return NULL; }
	const int64_t rows = (int64_t)t->n * t->h * t->w;
	if (t->prod >= 0) {
		MLOp *op = &C->ops[t->prod];
		t->sz32 = (size_t)rows * t->c * sizeof(float);
		t->d32 = (float*)dalloc_ex(C, t->sz32, 0, t->prod); t->ld32 = t->c;
		op->u.gemm.C32 = t->d32; op->u.gemm.ldc32 = t->c;
		return t->d32;
	}
	mlctx_fail(C, "tensor %s has no fp32 form", t->name);
	return NULL;
}

void* mlt_need16(MLCtx* C, MLTensor* t)
{
	if (!t) return NULL;
	if (t->d16) return t->d16;
	if (t->released) { mlctx_fail(C, "use of released tensor %s", t->name); return NULL; }
	const int64_t rows = (int64_t)t->n * t->h * t->w;
	if (t->prod >= 0) {
		MLOp *op = &C->ops[t->prod];
		if (t->c % 8) {
			/* consumer convs need 8-channel pixels: rows padded with zeros.  The GEMM never writes the pad
			 * columns, so the buffer is zeroed once here and never recycled (sz16 stays 0). */
			const int cpad = (t->c + 7) / 8 * 8;
			const size_t sz = (size_t)rows * cpad * 2;
			t->d16 = dalloc_ex(C, sz, 2, 0); t->ld16 = cpad;
			if (t->d16) mlsd_memset(t->d16, 0, sz, C->stream);
		} else {
			t->sz16 = (size_t)rows * t->c * 2;
			t->d16 = dalloc_ex(C, t->sz16, 0, t->prod); t->ld16 = t->c;
		}
		op = &C->ops[t->prod];
		op->u.gemm.C16 = t->d16; op->u.gemm.ldc16 = t->ld16;
		return t->d16;
	}
	if (t->is_input && t->in_type == MLT_F32 && t->h > 1) {
		/* image input in reference (NCHW fp32) layout -> channels-last fp16, channels padded to 8 */
		const int cpad = (t->c + 7) / 8 * 8;
		t->sz16 = (size_t)rows * cpad * 2;
		t->d16 = mlctx_dalloc(C, t->sz16, 0); t->ld16 = cpad;
		MLOp *op = mlctx_op_new(C, OP_NCHW2NHWC, "nchw_to_nhwc_f16");
		op->u.n2h.src = t->in_src ? t->in_src : (const float*)t->in_stage;
		op->u.n2h.n_src = t->in_src ? t->in_src_n : t->n; op->u.n2h.C = t->c; op->u.n2h.HW = t->h*t->w;
		op->u.n2h.dst = t->d16; op->u.n2h.n_dst = t->n; op->u.n2h.Cpad = cpad;
		op->u.n2h.scale = t->in_scale; op->u.n2h.scale0 = t->in_scale0; op->u.n2h.mode = t->in_mode;
		return t->d16;
	}
	if (t->d32 && t->ld32 == t->c) {
		t->sz16 = (size_t)rows * t->c * 2;
		t->d16 = mlctx_dalloc(C, t->sz16, 0); t->ld16 = t->c;
		MLOp *op = mlctx_op_new(C, OP_ACT, "f32_to_f16");
		op->u.act.x = t->d32; op->u.act.y = t->d16; op->u.act.n = (size_t)rows * t->c; op->u.act.act = MLSD_ACT_NONE;
		return t->d16;
	}
	mlctx_fail(C, "tensor %s has no fp16 form", t->name);
	return NULL;
}

MLB_API void mlb_release(MLCtx* C, MLTensor* t)
{
	if (!t || t->released || t->is_input) return;
	t->released = 1;
	if (t->sz32) mlctx_drelease(C, t->d32, t->sz32);
	if (t->sz16) mlctx_drelease(C, t->d16, t->sz16);
	if (t->sz_silu) mlctx_drelease(C, t->silu16, t->sz_silu);
	t->d32 = NULL; t->d16 = NULL; t->silu16 = NULL; t->sz32 = t->sz16 = t->sz_silu = 0;
	/* an un-consumed GEMM output keeps its op pointers: the op still writes to memory that may be reused
	 * only AFTER this point in the recorded order, which stream order makes safe. */
}

/* ------------------------------------------------------------------ naming records (src/mlblock.h:115-132) */
MLB_API void mlctx_block_begin(MLCtx* C)
{
	MLNameRec *r = VEC_PUSH(C, C->names, C->n_names, C->cap_names, MLNameRec);
	memset(r, 0, sizeof(*r)); r->param = -1;
}

void mlctx_named_op(MLCtx* C, const char* name)
{
	MLNameRec *r = VEC_PUSH(C, C->names, C->n_names, C->cap_names, MLNameRec);
	memset(r, 0, sizeof(*r)); r->kind = 1; r->name = strdup(name); r->param = -1;
}

MLB_API MLTensor* mlctx_tensor_add(MLCtx* C, const char* name, MLTensor* t)
{
	mlctx_named_op(C, name);
	if (t) {
		snprintf(t->name, sizeof(t->name), "%s", name); C->result = t;
		memcpy(C->names[C->n_names-1].ne, t->ne, sizeof(t->ne));
	}
	return t;
}

/* mlctx_build_alloc (src/mlblock.h:103, src/mlblock.c:161-176): finish the plan for `result` without loading parameters */
MLB_API int mlctx_build_alloc(MLCtx* C, MLTensor* result)
{
	if (result) C->result = result;
	return mlctx_prep(C);
}

/* mlctx_block_graph_dump(_path) (src/mlblock.c:347-388, MLIS_DUMP_GRAPH): the block tree, one line per named tensor, indented by
 * scope, walked backwards like the name resolution; "name: KIND TYPE [ne0,ne1,ne2,ne3]" with KIND = PARAM or BLOCK (a named
 * block output; the reference prints the ggml op that produced it, here a block is a group of fused launches). */
MLB_API int mlctx_block_graph_dump_path(const MLCtx* C, const char* path)
{
	FILE *f = fopen(path, "w");
	if (!f) return mlsd_set_error(-1, "could not open '%s'", path);
	int depth = 0, R = 1;
	for (int i=C->n_names-1; i>=0; --i) {
		const MLNameRec *r = &C->names[i];
		if (r->kind == 0) {
			if (!depth) { fputs("ERROR INVALID ML BLOCK GRAPH\n", f); R = -1; break; }
			--depth;
			continue;
		}
		for (int k=0;k<depth;++k) fputs("  ", f);
		if (r->kind == 2) {
			const MLParam *p = &C->params[r->param];
			fprintf(f, "%s: PARAM %s [%lld,%lld,%lld,%lld]\n", r->name, p->type == MLT_F16 ? "f16" : "f32",
				(long long)p->ne[0], (long long)p->ne[1], (long long)p->ne[2], (long long)p->ne[3]);
		} else {
			fprintf(f, "%s: BLOCK f32 [%lld,%lld,%lld,%lld]\n", r->name, (long long)r->ne[0], (long long)r->ne[1], (long long)r->ne[2], (long long)r->ne[3]);
			++depth;
		}
	}
	fclose(f);
	return R;
}

MLParam* mlctx_param_new(MLCtx* C, const char* name, int type, int64_t n0, int64_t n1, int64_t n2, int64_t n3,
	int layout, int64_t lp0, int64_t lp1)
{
	(void)lp0; (void)lp1;
	return mlctx_param_new_at(C, name, type, n0, n1, n2, n3, layout, NULL);
}

MLParam* mlctx_param_new_at(MLCtx* C, const char* name, int type, int64_t n0, int64_t n1, int64_t n2, int64_t n3,
	int layout, void* dev)
{
	MLParam *p = VEC_PUSH(C, C->params, C->n_params, C->cap_params, MLParam);
	memset(p, 0, sizeof(*p));
	p->type = type; p->ne[0]=n0; p->ne[1]=n1; p->ne[2]=n2; p->ne[3]=n3; p->layout = layout;
	size_t elems = (size_t)(n0*n1*n2*n3);
	if (layout == 1) {        /* conv [k0,k1,cin,cout] -> [cout][k1][k0][cin_pad] */
		const int64_t cpad = (n2 + 7) / 8 * 8;
		p->lp[0]=n0; p->lp[1]=n1; p->lp[2]=n2; p->lp[3]=n3; p->lp[4]=cpad;
		elems = (size_t)(n0*n1*cpad*n3);
	} else if (layout == 2) { /* GEGLU weight [n_in, 2d] */
		p->lp[0]=n0; p->lp[1]=n1/2;
	} else if (layout == 3) { /* GEGLU bias [2d] */
		p->lp[0]=1; p->lp[1]=n0/2;
	}
	p->dev_elems = elems;
	const size_t esz = type == MLT_F16 ? 2 : 4;
	p->dev = dev ? dev : mlctx_walloc(C, elems * esz);
	if (p->dev && layout == 1 && p->lp[4] != n2 && !is_virtual(C, p->dev)) mlsd_memset(p->dev, 0, elems * esz, C->stream);   /* (the host master copy of streamed weights starts zeroed) */
	MLNameRec *r = VEC_PUSH(C, C->names, C->n_names, C->cap_names, MLNameRec);
	r->kind = 2; r->name = strdup(name); r->param = C->n_params - 1;
	return &C->params[C->n_params - 1];
}

/* mlctx_load_prep, src/mlblock.c:67-105: walk the records backwards; a named op opens a scope that the
 * matching BLOCK_BEGIN closes; a parameter's key is the scope path + its own name. */
static int resolve_names(MLCtx* C)
{
	char path[512] = "";
	int stack[256], sp = 0;
	if (C->tprefix[0]) mlctx_named_op(C, C->tprefix);   /* mlctx_prep :318 */
	for (int i=C->n_names-1; i>=0; --i) {
		MLNameRec *r = &C->names[i];
		int nlen = (int)strlen(path);
		if (r->kind == 0) {
			if (!sp) return mlctx_fail(C, "invalid ML graph (unbalanced block)");
			path[stack[--sp]] = 0;
		} else {
			if (nlen + strlen(r->name) + 2 >= sizeof(path)) return mlctx_fail(C, "tensor name too long");
			snprintf(path + nlen, sizeof(path) - nlen, "%s%s", nlen ? "." : "", r->name);
			if (r->kind == 2) {
				free(C->params[r->param].key);
				C->params[r->param].key = strdup(path);
				path[nlen] = 0;
			} else {
				if (sp == 256) return mlctx_fail(C, "block nesting too deep");
				stack[sp++] = nlen;
			}
		}
	}
	return 1;
}

/* ------------------------------------------------------------------ inputs */
static MLTensor* input_finish(MLCtx* C, MLTensor* t, const char* name, int dtype)
{
	t->is_input = 1; t->in_type = dtype; t->in_scale0 = 1.0f; t->def_op = -1;
	snprintf(t->name, sizeof(t->name), "%s", name);
	t->in_bytes = (size_t)(t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]) * 4;
	t->in_stage = mlctx_dalloc(C, t->in_bytes, 0);
	if (dtype == MLT_F32 && t->h == 1) { t->d32 = (float*)t->in_stage; t->ld32 = t->c; }
	*VEC_PUSH(C, C->inputs, C->n_inputs, C->cap_inputs, MLTensor*) = t;
	return t;
}

MLTensor* mlctx_input_new_seq(MLCtx* C, const char* name, int dtype, int d, int T, int N)
{	/* [d,T,N,1]: channels-last storage equals the reference order */
	MLTensor *t = mlt_new(C, N, 1, T, d);
	t->ne[0]=d; t->ne[1]=T; t->ne[2]=N; t->ne[3]=1;
	return input_finish(C, t, name, dtype);
}

MLTensor* mlctx_input_new_img(MLCtx* C, const char* name, int w, int h, int c, int n)
{	/* [W,H,C,N]: host/staging data is NCHW fp32 (LocalTensor, src/localtensor.h:16-20) */
	MLTensor *t = mlt_new(C, n, h, w, c);
	t->ne[0]=w; t->ne[1]=h; t->ne[2]=c; t->ne[3]=n;
	if (h == 1) { t->h = 1; }
	return input_finish(C, t, name, MLT_F32);
}

MLB_API MLTensor* mlctx_input_new(MLCtx* C, const char* name, int dtype, int n0, int n1, int n2, int n3)
{	/* reference signature (src/mlblock.h:134-143).  [W,H,C,N] images have a small channel dim and H>1;
	 * everything else is a sequence/vector [d,T,N]. */
	if (dtype == MLT_F32 && (n3 > 1 || (n2 > 1 && n2 <= 16 && n1 > 1 && n0 > 1))) return mlctx_input_new_img(C, name, n0, n1, n2, n3);
	if (n3 > 1) { mlctx_fail(C, "mlctx_input_new(%s): unsupported 4-D non-image input", name); return NULL; }
	return mlctx_input_new_seq(C, name, dtype, n0, n1, n2);
}

MLB_API int mlctx_input_bind(MLTensor* t, const float* dev_src, int n_src, const float* dev_scale, float scale0, int mode)
{
	if (!t || !t->is_input || t->d16) return -1;   /* must be bound before the first consumer is recorded */
	t->in_src = dev_src; t->in_src_n = n_src; t->in_scale = dev_scale; t->in_scale0 = scale0; t->in_mode = mode;
	return 1;
}

MLB_API void* mlctx_input_device_ptr(MLTensor* t)
{	/* whoever asks for the staging buffer is about to write it: step-invariant ops fed by it run again */
	if (t && t->dirty) *t->dirty = 0;
	return t ? t->in_stage : NULL;
}

MLB_API int mlctx_input_set(MLCtx* C, MLTensor* t, const void* host, size_t nbytes)
{
	if (!t || !t->is_input) return mlctx_fail(C, "mlctx_input_set: not an input tensor");
	if (nbytes != t->in_bytes) return mlctx_fail(C, "mlctx_input_set(%s): size %zu != %zu", t->name, nbytes, t->in_bytes);
	if (t->dirty) *t->dirty = 0;
	if (mlsd_memcpy(t->in_stage, host, nbytes, 0, C->stream)) return -1;
	if (mlsd_stream_sync(C->stream)) return -1;
	return 1;
}

/* ------------------------------------------------------------------ LocalTensor (src/localtensor.h) and the all-in-one run */
MLB_API size_t ltensor_nelements(const LocalTensor* S) { return (size_t)S->n[0] * S->n[1] * S->n[2] * S->n[3]; }
MLB_API size_t ltensor_nbytes(const LocalTensor* S) { return sizeof(float) * ltensor_nelements(S); }

MLB_API void ltensor_free(LocalTensor* S)
{
	if (!S) return;
	if (S->flags & LT_F_OWNMEM) free(S->d);
	memset(S, 0, sizeof(*S));
}

MLB_API void ltensor_resize(LocalTensor* S, int n0, int n1, int n2, int n3)
{
	float *keep = (S->flags & LT_F_OWNMEM) ? S->d : NULL;        /* borrowed memory is let go, never reallocated */
	S->n[0] = n0; S->n[1] = n1; S->n[2] = n2; S->n[3] = n3;
	const size_t nb = ltensor_nbytes(S);
	S->d = (float*)realloc(keep, nb ? nb : 4);
	S->flags |= LT_F_OWNMEM;
}

MLB_API int ltensor_shape_check(const LocalTensor* S, int n0, int n1, int n2, int n3)
{
	const int want[4] = { n0, n1, n2, n3 };
	for (int i=0;i<4;++i) if (want[i] > 0 && want[i] != S->n[i]) return -1;
	return 1;
}

MLB_API int ltensor_finite_check(const LocalTensor* S)
{
	const size_t n = ltensor_nelements(S);
	for (size_t i=0;i<n;++i) if (!isfinite(S->d[i])) return -1;
	return 1;
}

MLB_API int ltensor_to_backend(MLCtx* C, const LocalTensor* S, MLTensor* t)
{
	if (!S || !S->d || !t) return mlctx_fail(C, "ltensor_to_backend: null tensor");
	if (t->in_type != MLT_F32) return mlctx_fail(C, "ltensor_to_backend(%s): the plan input is not fp32", t->name);
	return mlctx_input_set(C, t, S->d, ltensor_nbytes(S));      /* (refuses a size mismatch: the reference asserts) */
}

MLB_API int ltensor_from_backend(MLCtx* C, LocalTensor* S, MLTensor* t)
{
	if (!S || !t) return mlctx_fail(C, "ltensor_from_backend: null tensor");
	if (!mlt_need32(C, t)) return -1;
	ltensor_resize(S, (int)t->ne[0], (int)t->ne[1], (int)t->ne[2], (int)t->ne[3]);
	return mlctx_output_get(C, t, S->d, ltensor_nbytes(S));
}

MLB_API void mlctx_set_tstore(MLCtx* C, const struct MLTStore* S) { C->tstore = S; }

int mlctx_tstore_load(MLCtx* C, const struct MLTStore* S);

MLB_API int mlctx_run_(MLCtx* C, LocalTensor* out, const LocalTensor** inputs)
{
	int R = 1;
	if (mlctx_prep(C) < 0) { R = -1; goto end; }
	if (C->tstore) { if (mlctx_tstore_load(C, C->tstore) < 0) { R = -1; goto end; } }
	else if (C->flags & MLB_F_SYNTH_PARAMS) { if (mlctx_params_synth(C, 1234) < 0) { R = -1; goto end; } }
	for (int i=0; i<C->n_inputs && inputs && inputs[i]; ++i)
		if (ltensor_to_backend(C, inputs[i], C->inputs[i]) < 0) { R = -1; goto end; }
	if (mlctx_compute(C) < 0) { R = -1; goto end; }
	if (out && ltensor_from_backend(C, out, C->result) < 0) { R = -1; goto end; }
end:
	mlctx_end(C);               /* the reference frees the context's graph and buffers here (mlctx_free: the MLCtx object itself stays usable) */
	return R;
}

MLB_API const float* mlctx_tensor_device_f32(MLCtx* C, MLTensor* t, int64_t* ld)
{
	(void)C;
	if (!t || !t->d32) return NULL;
	if (ld) *ld = t->ld32;
	return t->d32;
}

MLB_API int mlctx_output_get(MLCtx* C, MLTensor* t, float* out, size_t nbytes)
{
	if (!t || !t->d32) return mlctx_fail(C, "mlctx_output_get: tensor has no fp32 data");
	const int64_t rows = (int64_t)t->n * t->h * t->w, c = t->c;
	if (nbytes != (size_t)rows * c * 4) return mlctx_fail(C, "mlctx_output_get(%s): size %zu != %zu", t->name, nbytes, (size_t)rows*c*4);
	if (t->ld32 != c) return mlctx_fail(C, "mlctx_output_get: strided tensor");
	if (t->h == 1) {   /* sequence: channels-last == reference order */
		if (mlsd_memcpy(out, t->d32, nbytes, 1, C->stream) || mlsd_stream_sync(C->stream)) return -1;
		return 1;
	}
	float *tmp = (float*)malloc(nbytes);
	if (mlsd_memcpy(tmp, t->d32, nbytes, 1, C->stream) || mlsd_stream_sync(C->stream)) { free(tmp); return -1; }
	const int64_t hw = (int64_t)t->h * t->w;   /* [n][hw][c] -> [n][c][hw] (ltensor_from_backend gives NCHW) */
	for (int64_t n=0;n<t->n;++n) for (int64_t p=0;p<hw;++p) for (int64_t ch=0;ch<c;++ch)
		out[(n*c + ch)*hw + p] = tmp[(n*hw + p)*c + ch];
	free(tmp);
	return 1;
}

/* ------------------------------------------------------------------ prep / compute */
static int run_op(MLCtx* C, MLOp* op)
{
	void *st = C->stream;
	switch (op->kind) {
	case OP_GEMM: return mlsd_gemm(&op->u.gemm, st);
	case OP_ATTN: return mlsd_attention(&op->u.attn, st);
	case OP_GN:   if (op->fused) return 0;                    /* its producer's reduce pass ends with it (wire_gn_fold) */
	              return mlsd_groupnorm(&op->u.gn, st);
	case OP_LN:   if (op->fused) return 0;                    /* its producer ends with it (wire_ln_fold) */
	              return mlsd_layernorm(op->u.ln.x, op->u.ln.ldx, op->u.ln.rows, op->u.ln.d, op->u.ln.eps, op->u.ln.g, op->u.ln.b,
	                                   op->u.ln.y16, op->u.ln.y32, st);
	case OP_NCHW2NHWC: return mlsd_nchw_to_nhwc_f16(op->u.n2h.src, op->u.n2h.n_src, op->u.n2h.C, op->u.n2h.HW, op->u.n2h.dst,
	                                   op->u.n2h.n_dst, op->u.n2h.Cpad, op->u.n2h.scale, op->u.n2h.scale0, op->u.n2h.mode, st);
	case OP_NHWC2NCHW: return mlsd_nhwc_to_nchw_f32(op->u.h2n.src, op->u.h2n.ld, op->u.h2n.n, op->u.h2n.C, op->u.h2n.HW,
	                                   op->u.h2n.dst, op->u.h2n.mul, op->u.h2n.add, st);
	case OP_TEMB: return mlsd_timestep_embedding(op->u.temb.t, op->u.temb.n, op->u.temb.dim, op->u.temb.maxp, op->u.temb.out, st);
	case OP_ACT:  return mlsd_act_f32_to_f16(op->u.act.x, op->u.act.y, op->u.act.n, op->u.act.act, st);
	case OP_CLIP_EMBED: return mlsd_clip_embed(op->u.cemb.tok, op->u.cemb.n, op->u.cemb.T, op->u.cemb.d, op->u.cemb.tw,
	                                   op->u.cemb.pw, op->u.cemb.out, st);
	case OP_SOFTMAX: return mlsd_softmax_rows(op->u.smax.in, op->u.smax.ld_in, op->u.smax.out, op->u.smax.ld_out,
	                                   op->u.smax.rows, op->u.smax.cols, op->u.smax.scale, st);
	case OP_COPY_F32: return mlsd_memcpy(op->u.copy.dst, op->u.copy.src, op->u.copy.nbytes, 2, st);
	case OP_XA_VT: return mlsd_xattn_pack_vt(op->u.xavt.v, op->u.xavt.ldv, op->u.xavt.n_img, op->u.xavt.Tk, op->u.xavt.N, op->u.xavt.vt, st);
	}
	return mlsd_set_error(-1, "unknown op kind %d", (int)op->kind);
}

/* ------------------------------------------------------------------ GEMM tile selection
 * The best tile variant of a GEMM/conv depends on how its (M,N,K) grid quantises over the 256 CUs and on the reduction
 * length.  Selection is a PURE FUNCTION OF THE SHAPE: mlctx_prep looks every GEMM up in the table compiled in from
 * tune_table.inc (keyed by shape + epilogue form); a shape that is not listed takes the launcher's static choice with no
 * K split.  Two processes - or the ranks of a multi-GPU job - therefore always run the same kernels in the same fp32
 * summation order and produce bit-identical results (tests/test_determinism_gpu.py).
 * The table is produced OFFLINE by tools/tune_all.py: with MLSD_AUTOTUNE=1 (or mlctx_set_autotune(1)) the first
 * mlctx_compute of a plan times the candidate variants of every shape the table does not hold (HIP events on the plan's
 * stream, min of 3, on the op's real operands) and mlsd_tune_dump() writes the winners as table lines. */
typedef struct { int conv, M, N, K, act, OH, stride, ups, out, Cin, KH, rb; int best, ksplit; } TuneKey;
static const TuneKey k_tune_builtin[] = {
#include "tune_table.inc"
	{ -1, 0,0,0,0,0,0,0,0,0,0,0, 0,0 }
};
static TuneKey g_tune[1024];
static int g_ntune = 0;
static int g_autotune = -1;          /* -1: not decided yet (environment), 0: table only, 1: time unknown shapes */
static int g_tune_miss = 0;
#include <pthread.h>
static pthread_mutex_t g_tune_mu = PTHREAD_MUTEX_INITIALIZER;

static int autotune_on(void)
{
	if (g_autotune < 0) { const char *e = getenv("MLSD_AUTOTUNE"); g_autotune = (e && *e && *e != '0') ? 1 : 0; }
	return g_autotune;
}

MLB_API void mlctx_set_autotune(int on) { g_autotune = on ? 1 : 0; }
MLB_API int mlctx_tune_misses(void) { return g_tune_miss; }
MLB_API int mlctx_plan_tune_misses(const MLCtx* C) { return C ? C->n_tune_miss : 0; }   /* GEMM shapes of THIS plan the table does not list */
MLB_API int mlctx_plan_tune_nearest(const MLCtx* C) { return C ? C->n_tune_near : 0; }   /* ... of which a table entry with the nearest row count lent its tile (the rest: static rule) */

static TuneKey tune_key(const mlsd_gemm_args* g)
{
	TuneKey k = { g->conv, g->M, g->N, g->K, g->act, g->conv ? g->OH : 0, g->conv ? g->stride : 0, g->conv ? g->upsample : 0,
		(g->C32 ? 1 : 0) | (g->C16 ? 2 : 0) | (g->resid ? 4 : 0) | (g->bias_m ? 8 : 0) | (g->act_after_resid ? 16 : 0),
		g->conv ? g->Cin : 0, g->conv ? g->KH : 0, g->rowbias ? 1 : 0, 0, 0 };
	return k;
}

static int tune_eq(const TuneKey* t, const TuneKey* k)
{
	return t->conv==k->conv && t->M==k->M && t->N==k->N && t->K==k->K && t->act==k->act && t->OH==k->OH && t->stride==k->stride &&
	       t->ups==k->ups && t->out==k->out && t->Cin==k->Cin && t->KH==k->KH && t->rb==k->rb;
}

/* table lookup: the process cache (filled by the timing mode) first, then the compiled-in table */
static int tune_lookup(const TuneKey* k, int* best, int* ksplit)
{
	int found = 0;
	pthread_mutex_lock(&g_tune_mu);
	for (int i=0; i<g_ntune && !found; ++i) if (tune_eq(&g_tune[i], k)) { *best = g_tune[i].best; *ksplit = g_tune[i].ksplit; found = 1; }
	pthread_mutex_unlock(&g_tune_mu);
	static int ignore_builtin = -1;     /* tools/tune_all.py --fresh: re-time every shape */
	if (ignore_builtin < 0) { const char *e = getenv("MLSD_TUNE_IGNORE_TABLE"); ignore_builtin = (e && *e && *e != '0') ? 1 : 0; }
	for (const TuneKey *t = k_tune_builtin; !found && !ignore_builtin && t->conv >= 0; ++t)
		if (tune_eq(t, k)) { *best = t->best; *ksplit = t->ksplit; found = 1; }
	return found;
}

/* NEAREST-SHAPE lookup (round 5): the table is keyed on exact shapes, and a size nobody tuned (SDXL 768 x 768, 1024 x 768: every GEMM has another M) fell to the static
 * rule -- measured 1.9 x the tuned plan's time per FLOP (tests/test_unet_gpu.py::test_unet_sizes_nobody_tuned_parity_and_speed).  The tile that suits a problem depends on
 * N, K, the epilogue kind and the MAGNITUDE of M: on an exact miss the entry that agrees in everything but the row count (and the image height that goes with it) and whose
 * M is nearest on a log scale (within 4 x) lends its variant and K split.  Still a pure function of the shape: plans stay deterministic. */
static int g_nearest_off = -1;
MLB_API void mlctx_set_nearest_tile(int on) { g_nearest_off = on ? 0 : 1; }      /* A/B: 0 = exact table hits only (misses take the static rule), as before round 5 */
static int tune_lookup_nearest(const TuneKey* k, int* best, int* ksplit)
{
	if (g_nearest_off < 0) { const char *e = getenv("MLSD_NO_NEAREST_TILE"); g_nearest_off = (e && *e && *e != '0') ? 1 : 0; }
	/* (M >= 1024: plans whose largest launches are smaller than that -- test-sized latents -- are bound by their dispatch count whatever the tile, and the entries nearest to
	 * them belong to SD1.5 batch 1's 8 x 8 level, with K splits 20-45 deep) */
	if (g_nearest_off || k->M < 1024) return 0;
	double dbest = 1e30; int found = 0;
	for (int pass=0; pass<2; ++pass) {
		const TuneKey *t = pass ? k_tune_builtin : g_tune;
		const int n = pass ? (int)(sizeof(k_tune_builtin) / sizeof(k_tune_builtin[0])) - 1 : g_ntune;
		if (!pass) pthread_mutex_lock(&g_tune_mu);
		for (int i=0;i<n;++i) {
			const TuneKey *q = &t[i];
			if (q->conv != k->conv || q->N != k->N || q->K != k->K || q->act != k->act || q->stride != k->stride || q->ups != k->ups || q->out != k->out ||
			    q->Cin != k->Cin || q->KH != k->KH || q->rb != k->rb || q->M <= 0) continue;
			double d = log((double)q->M / (double)k->M); if (d < 0) d = -d;
			if (d > 1.3863 /* ln 4 */) continue;
			if (d < dbest || (d == dbest && q->M > k->M)) { dbest = d; *best = q->best; *ksplit = q->ksplit; found = 1; }
		}
		if (!pass) pthread_mutex_unlock(&g_tune_mu);
	}
	return found;
}

/* writes the process cache (shapes timed in this process) as lines of tune_table.inc; returns the number written */
MLB_API int mlsd_tune_dump(const char* path)
{
	FILE *f = fopen(path, "w");
	if (!f) return mlsd_set_error(-1, "mlsd_tune_dump: cannot open %s", path);
	pthread_mutex_lock(&g_tune_mu);
	for (int i=0;i<g_ntune;++i) {
		const TuneKey *t = &g_tune[i];
		fprintf(f, "\t{ %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d },\n", t->conv, t->M, t->N, t->K, t->act, t->OH, t->stride,
			t->ups, t->out, t->Cin, t->KH, t->rb, t->best, t->ksplit);
	}
	const int n = g_ntune;
	pthread_mutex_unlock(&g_tune_mu);
	fclose(f);
	return n;
}

#define SPLITK_WS_BYTES ((size_t)128 << 20)

static int splitk_ws_get(MLCtx* C, mlsd_gemm_args* g)
{
	if (!C->splitk_ws) {
		if (mlsd_malloc(&C->splitk_ws, SPLITK_WS_BYTES)) return -1;
		C->splitk_ws_bytes = SPLITK_WS_BYTES;
		C->mem_compute += SPLITK_WS_BYTES;
	}
	g->ws = C->splitk_ws; g->ws_bytes = C->splitk_ws_bytes;
	return 0;
}

static int streamk_get(MLCtx* C, mlsd_gemm_args* g)
{	/* slabs in the split-K workspace (ops run one at a time on the plan's stream) + the plan's flag words (zero between launches:
	 * the kernels that use them clear them again) */
	if (splitk_ws_get(C, g)) return -1;
	if (!C->sk_flags) {
		if (mlsd_malloc((void**)&C->sk_flags, SK_FLAG_WORDS * 4)) return -1;
		if (mlsd_memset(C->sk_flags, 0, SK_FLAG_WORDS * 4, C->stream) || mlsd_stream_sync(C->stream)) return -1;
	}
	g->sk_flags = C->sk_flags;
	return 0;
}

/* prep-time selection: table hit -> its variant / K split; miss -> the launcher's static choice (tile_variant 0) */
static int select_gemm(MLCtx* C, MLOp* op)
{
	mlsd_gemm_args *g = &op->u.gemm;
	if (g->xa_k && mlsd_gemm_xattn_fused(g) == 1) { g->tile_variant = 19; g->ksplit = 1; return 1; }      /* a q projection that ends with its cross attention: one tile does that (a shape rule above the table, not a miss) */
	if (mlsd_conv_smalln_eligible(g)) { g->tile_variant = 0; g->ksplit = 1; return 1; }                       /* likewise the small-Cout streaming convolution */
	const TuneKey k = tune_key(g);
	int best = 0, ks = 1;
	int exact = tune_lookup(&k, &best, &ks);
	if (!exact && tune_lookup_nearest(&k, &best, &ks)) { exact = 2; C->n_tune_near++; }
	if (exact) {
		{	/* MLSD_NO_TT=1 (A/B, debugging): table entries of the 128 x 160 kernel run on the 128 x 320 ping-pong tile */
			static int no_tt = -1;
			if (no_tt < 0) { const char *e = getenv("MLSD_NO_TT"); no_tt = (e && *e && *e != '0') ? 1 : 0; }
			if (no_tt && best == VARIANT_TT) best = 19;
		}
		/* a plan on a CU-masked stream (mlctx_set_cus): no in-launch hand-offs -- a stream-K entry runs as the plain tile of its shape */
		if (IS_STREAMK(best) && C->cu_budget > 0 && C->cu_budget < 256) best = best == VARIANT_STREAMK ? 18 : 19;
		g->tile_variant = best; g->ksplit = ks;
		if (ks > 1 && (size_t)ks * g->M * g->N * sizeof(float) > SPLITK_WS_BYTES) g->ksplit = 1;
		if (g->ksplit > 1 && streamk_get(C, g)) return -1;
		if ((IS_STREAMK(best) || best == VARIANT_SKINNY) && streamk_get(C, g)) return -1;
		/* The 128 x 160 two-tiles-per-CU kernel (gemm_tt.hip, round 5) instead of the 128 x 320 ping-pong tile WHERE THAT TILE FILLS AT MOST HALF OF THE CUs (a pure function
		 * of the shape).  In-plan A/B, same box (profiles/r5_gemm_tt_inplan.txt): SDXL b2 evaluation -2.4 %, b1 -0.9 %, SD1.5 b2 -1.1 %; on the full-chip single-round
		 * launches of SDXL b4 it LOSES (+0.4 .. +0.7 %: its loop is 25 % slower than the ping-pong loop, which eats the overlap it buys) and is not used there.
		 * MLSD_TT: 0 = never, 1 = also where K >= 2560, 2 = on every shape it takes (the A/B settings of that study), 3 / unset = the rule above. */
		static int tt_mode = -1;
		if (tt_mode < 0) { const char *e = getenv("MLSD_TT"); tt_mode = e && *e ? atoi(e) : 3; }
		if (tt_mode > 0 && (best == 19 || best == 21) && !g->conv && g->act == MLSD_ACT_NONE && !(g->M % 128) && !(g->N % 160) && !(g->K & 63) && g->ksplit <= 1 &&
		    !g->rowbias && !g->bias_m && (tt_mode == 2 || (tt_mode == 1 && g->K >= 2560) || (tt_mode >= 1 && (long)(g->M / 128) * ((g->N + 319) / 320) <= 128)) &&
		    !(C->cu_budget > 0 && C->cu_budget < 256)) g->tile_variant = VARIANT_TT;
		return exact;
	}
	g->tile_variant = 0; g->ksplit = 1;
	return 0;
}

static int time_gemm(MLCtx* C, mlsd_gemm_args* g, void* e0, void* e1, float* ms_out)
{
	if (mlsd_gemm(g, C->stream)) return -1;   /* warm-up */
	float mn = 1e30f;
	for (int r=0;r<3;++r) {
		mlsd_event_record(e0, C->stream);
		if (mlsd_gemm(g, C->stream)) return -1;
		mlsd_event_record(e1, C->stream);
		mlsd_event_sync(e1);
		float ms = 0; mlsd_event_elapsed_ms(e0, e1, &ms);
		if (ms < mn) mn = ms;
	}
	*ms_out = mn;
	return 0;
}

/* candidates of one GEMM: (tile variant, K slices), in order of preference; returns their number (<= 32) */
static int gemm_candidates(MLCtx* C, mlsd_gemm_args* g, int cv[32], int cs[32])
{
	int nc = 0;
	/* skinny-M weight streaming (variant 29): K slices so that (N / 64) x slices is about one / two blocks per CU */
	if (g->M <= 128 && !(g->K & 63) && g->K >= 256 && g->act != MLSD_ACT_GEGLU && !(g->N & 3) && (!g->conv || (!g->upsample && !(g->Cin & 63))) && !streamk_get(C, g)) {
		const int nb = (g->N + 63) / 64, nkt = g->K / 64;
		int last = 0;
		for (int t = 256; t <= 1024 && nc < 6; t *= 2) {
			int s = (t + nb / 2) / nb;
			if (s < 1) s = 1;
			if (s > nkt / 2) s = nkt / 2;
			while (s > 1 && (size_t)s * g->M * g->N * sizeof(float) > SPLITK_WS_BYTES) --s;
			if (s == last) continue;
			cv[nc] = 29; cs[nc++] = s; last = s;
		}
	}
	if (g->M <= 64) { cv[nc]=1; cs[nc++]=1; cv[nc]=0; cs[nc++]=1; }
	else {
		/* persistent ping-pong tiles (gemm_pp.hpp): problems made of whole wave blocks; convs whose K tiles lie inside
		 * one filter tap (anything else they would hand to the LDS-transposing tile of the same shape anyway) */
		const int pp_ok = !(g->K & 63) && g->K >= 192 && (!g->conv || (!(g->Cin & 63) && (!g->upsample || (g->stride == 1 && g->C32 && !g->C16 && !g->resid && g->act == MLSD_ACT_NONE))));      /* (upsampled sources: round 5, gemm_pp.hpp CONV == 2) */
		if (pp_ok && g->M >= 256 && g->N >= 256 && !(g->M & 127) && !(g->N & 63)) {
			cv[nc]=17; cs[nc++]=1; cv[nc]=21; cs[nc++]=1;
			/* stream-K on the same tile where the tiles do not fill whole rounds of the 256 blocks */
			const long t256 = (long)((g->M + 255) / 256) * ((g->N + 255) / 256);
			if (t256 % 256 && t256 * (g->K / 64) >= 256 * 4 && g->act != MLSD_ACT_GEGLU && !(g->conv && g->upsample) && !streamk_get(C, g)) { cv[nc]=19; cs[nc++]=1; }
		}
		if (pp_ok && g->M >= 128 && !(g->M & 63) && !(g->N % 80) && g->act != MLSD_ACT_GEGLU) {
			cv[nc]=18; cs[nc++]=1; cv[nc]=20; cs[nc++]=1;   /* four / two phases per K tile */
			/* stream-K on that tile: few tiles (not whole rounds of the 256 blocks) and K long enough to deal out */
			const long t320 = (long)((g->M + 127) / 128) * ((g->N + 319) / 320);
			if (t320 % 256 && t320 < 256 && t320 * (g->K / 64) >= 256 * 4 && !(g->conv && g->upsample) && !streamk_get(C, g)) { cv[nc]=28; cs[nc++]=1; }
		}
		/* two tiles in flight per CU (gemm_tt.hip, round 5): linear / 1x1 problems made of whole 128 x 160 tiles */
		if (!(g->M % 128) && !(g->N % 160) && !(g->K & 63) && g->K >= 128 && g->act == MLSD_ACT_NONE && !g->rowbias && !g->bias_m &&
		    (!g->conv || (g->KH == 1 && g->KW == 1 && g->stride == 1 && g->pad == 0 && !g->upsample)) && !(g->C32 && g->C16)) { cv[nc]=30; cs[nc++]=1; }
		/* (the narrow 256x128 ping-pong tile, variant 25, is NOT a candidate: it wins this warm, back-to-back timing on the VAE's N = 128
		 * convolutions (+9..16 %) and loses in the plan (-16 %, profiles/r3_gemm_narrow_tile.txt): those launches are bound by their fp32
		 * output + residual traffic, which two co-resident blocks overlap with each other's K loops and one persistent block cannot) */
		cv[nc]=9; cs[nc++]=1; cv[nc]=3; cs[nc++]=1; cv[nc]=4; cs[nc++]=1; cv[nc]=0; cs[nc++]=1;
	}
	const long t128 = (long)((g->M + 127) / 128) * ((g->N + 127) / 128);
	if (g->M > 64 && t128 < 512) { cv[nc]=1; cs[nc++]=1; }
	/* 128x320: outputs whose width is a multiple of 320 (N = 1280, 640) in whole tile columns, e.g. 8192x1280 =
	 * 256 tiles = one per CU where 256x256 tiles give 160 */
	if (g->M > 64 && g->N >= 320 && g->act != MLSD_ACT_GEGLU) { cv[nc]=16; cs[nc++]=1; }   /* short problems: more, smaller blocks hide the fill latency */
	/* split-K: output tiles alone cannot occupy the 256 CUs (2 resident blocks each) and K is long enough to slice */
	const int nkt = (g->K + 63) / 64;
	if (t128 <= 192 && nkt >= 8 && g->act != MLSD_ACT_GEGLU && !(g->N & 3)) {
		static const int targets[] = {256, 512, 1024};
		for (int vi=0; vi<2; ++vi) {
			const int v = vi ? 1 : 0;                       /* 128x128 and 64x128 tiles */
			if (v == 1 && g->M > 512) continue;
			const long tiles = (long)((g->M + (v ? 63 : 127)) / (v ? 64 : 128)) * ((g->N + 127) / 128);
			int last = 1;
			for (int ti=0; ti<3; ++ti) {
				int s = (int)((targets[ti] + tiles / 2) / tiles);
				if (s > nkt / 2) s = nkt / 2;               /* >= 2 K tiles per slice */
				while (s > 1 && (size_t)s * g->M * g->N * sizeof(float) > SPLITK_WS_BYTES) --s;
				if (s <= last || nc >= 32) continue;
				cv[nc] = v; cs[nc++] = s; last = s;
			}
		}
	}
	return nc;
}

/* offline timing mode (tools/tune_all.py): called for shapes the table does not hold */
static int tune_gemm(MLCtx* C, MLOp* op)
{
	mlsd_gemm_args *g = &op->u.gemm;
	TuneKey k = tune_key(g);
	{
		int best = 0, ks = 1;
		if (tune_lookup(&k, &best, &ks)) {
			g->tile_variant = best; g->ksplit = ks;
			if (g->ksplit > 1 && streamk_get(C, g)) return -1;
			if ((IS_STREAMK(best) || best == VARIANT_SKINNY) && streamk_get(C, g)) return -1;
			return 1;
		}
	}
	/* candidates: (tile variant, K slices) */
	int cv[32], cs[32];
	const int nc = gemm_candidates(C, g, cv, cs);
	void *e0 = NULL, *e1 = NULL;
	if (mlsd_event_create(&e0) || mlsd_event_create(&e1)) return -1;
	float best_ms = 1e30f; int best = 0, best_s = 1;
	for (int c=0;c<nc;++c) {
		g->tile_variant = cv[c] + 1; g->ksplit = cs[c];
		float ms;
		if ((cs[c] > 1 && streamk_get(C, g)) || time_gemm(C, g, e0, e1, &ms)) { mlsd_event_destroy(e0); mlsd_event_destroy(e1); return -1; }
		/* candidates are listed in order of preference: a later one must win by 3 % (keeps the choice, and with
		 * it the kernel mix of a profile, stable against timing noise between near-equal variants) */
		if (ms < (c ? 0.97f : 1.f) * best_ms) { best_ms = ms; best = cv[c] + 1; best_s = cs[c]; }
	}
	mlsd_event_destroy(e0); mlsd_event_destroy(e1);
	g->tile_variant = best; g->ksplit = best_s;
	k.best = best; k.ksplit = best_s;
	pthread_mutex_lock(&g_tune_mu);
	if (g_ntune < 1024) g_tune[g_ntune++] = k;
	pthread_mutex_unlock(&g_tune_mu);
	return 1;
}


/* IN-PLAN tile tuning (round 3; tools/tune_inplan.py).  tune_gemm() above times a candidate back to back on warm operands; in the plan a launch
 * meets cold weights, a residual that comes from HBM and whatever clock the previous launches left, and round 3 measured tiles that win the
 * warm timing by 7-16 % and lose in the plan (profiles/r3_gemm_narrow_tile.txt, r3_gemm_w4.txt).  This pass times the candidates WHERE THEY RUN:
 * round r puts candidate r of every GEMM shape into all launches of that shape and times the whole plan op by op (HIP events, `reps` passes, the
 * smallest per-shape sum counts); a challenger replaces the current choice of a shape when it wins by 3 %.  Launches whose column statistics a
 * GroupNorm was wired to keep their tile.  Winners go to the process table (mlsd_tune_dump writes them).  Returns the number of shapes changed. */
MLB_API int mlctx_tune_inplan(MLCtx* C, int reps)
{
	if (!C->prepared) return mlctx_fail(C, "mlctx_tune_inplan before mlctx_prep");
	enum { MAXS = 256 };
	typedef struct { TuneKey k; int cv[33], cs[33], nc; double t[33]; } Shape;
	Shape *sh = calloc(MAXS, sizeof(Shape));
	int *op_shape = malloc(sizeof(int) * (size_t)C->n_ops);
	float *ms = malloc(sizeof(float) * (size_t)C->n_ops);
	if (!sh || !op_shape || !ms) { free(sh); free(op_shape); free(ms); return -1; }
	int ns = 0, maxnc = 0, changed = -1;
	for (int i=0;i<C->n_ops;++i) {
		op_shape[i] = -1;
		MLOp *op = &C->ops[i];
		if (op->kind != OP_GEMM || op->fused || op->u.gemm.colstats || op->u.gemm.ln_y16 || op->u.gemm.gn_y16) continue;     /* (tiles bound by a consumer's wiring stay) */
		mlsd_gemm_args *g = &op->u.gemm;
		const TuneKey k = tune_key(g);
		int j = 0;
		for (; j<ns; ++j) if (!memcmp(&sh[j].k, &k, offsetof(TuneKey, best))) break;
		if (j == ns) {
			if (ns == MAXS) continue;
			sh[j].k = k;
			sh[j].cv[0] = g->tile_variant - 1; sh[j].cs[0] = g->ksplit > 0 ? g->ksplit : 1;     /* candidate 0: the current choice (-1: the static rule) */
			int cv[32], cs[32];
			const int nc = gemm_candidates(C, g, cv, cs);
			sh[j].nc = 1;
			for (int c=0;c<nc;++c) if (cv[c] != sh[j].cv[0] || cs[c] != sh[j].cs[0]) { sh[j].cv[sh[j].nc] = cv[c]; sh[j].cs[sh[j].nc] = cs[c]; sh[j].nc++; }
			if (sh[j].nc > maxnc) maxnc = sh[j].nc;
			++ns;
		}
		op_shape[i] = j;
	}
	for (int r=0; r<maxnc; ++r) {
		for (int i=0;i<C->n_ops;++i) {
			const int j = op_shape[i];
			if (j < 0) continue;
			const int c = r < sh[j].nc ? r : 0;
			mlsd_gemm_args *g = &C->ops[i].u.gemm;
			g->tile_variant = sh[j].cv[c] + 1; g->ksplit = sh[j].cs[c];
			if ((g->ksplit > 1 || IS_STREAMK(g->tile_variant) || g->tile_variant == VARIANT_SKINNY) && streamk_get(C, g)) goto done;
		}
		for (int j=0;j<ns;++j) if (r < sh[j].nc) sh[j].t[r] = 1e30;
		for (int rep=0; rep<reps+1; ++rep) {             /* (the first pass warms the instruction caches of this round's kernels) */
			if (mlctx_profile_ops(C, ms, C->n_ops) < 0) goto done;
			if (!rep) continue;
			double sum[MAXS] = {0};
			for (int i=0;i<C->n_ops;++i) if (op_shape[i] >= 0) sum[op_shape[i]] += ms[i];
			for (int j=0;j<ns;++j) if (r < sh[j].nc && sum[j] < sh[j].t[r]) sh[j].t[r] = sum[j];
		}
	}
	changed = 0;
	for (int j=0;j<ns;++j) {
		int best = 0;
		for (int c=1;c<sh[j].nc;++c) if (sh[j].t[c] < 0.97 * sh[j].t[0] && sh[j].t[c] < sh[j].t[best]) best = c;
		sh[j].k.best = sh[j].cv[best] + 1; sh[j].k.ksplit = sh[j].cs[best];
		if (best) {
			++changed;
			pthread_mutex_lock(&g_tune_mu);
			int q = 0;
			for (; q<g_ntune; ++q) if (!memcmp(&g_tune[q], &sh[j].k, offsetof(TuneKey, best))) break;
			if (q < 1024) { g_tune[q] = sh[j].k; if (q == g_ntune) ++g_ntune; }
			pthread_mutex_unlock(&g_tune_mu);
			fprintf(stderr, "[tune_inplan] %s M=%d N=%d K=%d out=%d: variant %d k/%d %.3f ms -> variant %d k/%d %.3f ms per evaluation\n", sh[j].k.conv ? "conv" : "linear",
				sh[j].k.M, sh[j].k.N, sh[j].k.K, sh[j].k.out, sh[j].cv[0], sh[j].cs[0], sh[j].t[0], sh[j].cv[best], sh[j].cs[best], sh[j].t[best]);
		}
	}
	for (int i=0;i<C->n_ops;++i) {
		const int j = op_shape[i];
		if (j < 0) continue;
		mlsd_gemm_args *g = &C->ops[i].u.gemm;
		g->tile_variant = sh[j].k.best; g->ksplit = sh[j].k.ksplit;
		if ((g->ksplit > 1 || IS_STREAMK(g->tile_variant) || g->tile_variant == VARIANT_SKINNY) && streamk_get(C, g)) { changed = -1; break; }
	}
done:
	free(sh); free(op_shape); free(ms);
	return changed;
}

/* GroupNorm statistics from the producers: when every source of a GroupNorm is the fp32 output of a
 * ping-pong GEMM / conv launch that can emit column statistics (mlsd_gemm_colstats_rows > 0: fp32 output, no activation),
 * that launch gets a statistics buffer and the GroupNorm's first pass over the fp32 map is replaced by gn_finalize over the
 * statistics.  The choice is a pure function of the plan (tile table + shapes); MLSD_GN_TWO_PASS=1 keeps the two-pass form
 * (A/B timing, and the reference for the parity test). */
/* every buffer an op writes (at most 3) */
static int op_outputs(const MLOp* o, const void* out[3])
{
	int n = 0;
	switch (o->kind) {
	case OP_GEMM:
		if (o->u.gemm.C32) out[n++] = o->u.gemm.C32;
		if (o->u.gemm.C16) out[n++] = o->u.gemm.C16;
		if (o->u.gemm.ln_y16 && n < 3) out[n++] = o->u.gemm.ln_y16;
		if (o->u.gemm.gn_y16 && n < 3) out[n++] = o->u.gemm.gn_y16;
		if (o->u.gemm.xa_k && n < 3) out[n++] = o->u.gemm.xa_out;
		break;
	case OP_ATTN: out[n++] = o->u.attn.out; break;
	case OP_GN: out[n++] = o->u.gn.y16; if (o->u.gn.raw16) out[n++] = o->u.gn.raw16; break;
	case OP_LN: if (o->u.ln.y16) out[n++] = o->u.ln.y16; if (o->u.ln.y32) out[n++] = o->u.ln.y32; break;
	case OP_NCHW2NHWC: out[n++] = o->u.n2h.dst; break;
	case OP_NHWC2NCHW: out[n++] = o->u.h2n.dst; break;
	case OP_TEMB: out[n++] = o->u.temb.out; break;
	case OP_ACT: out[n++] = o->u.act.y; break;
	case OP_CLIP_EMBED: out[n++] = o->u.cemb.out; break;
	case OP_SOFTMAX: out[n++] = o->u.smax.out; break;
	case OP_COPY_F32: out[n++] = o->u.copy.dst; break;
	case OP_XA_VT: out[n++] = o->u.xavt.vt; break;
	}
	return n;
}

/* The producer of a GroupNorm source is the op that DEFINES the source tensor (MLTensor.prod, recorded with the GroupNorm:
 * MLOp.gn_src), not "the last GEMM that happens to write this address": arena blocks are recycled, so an address match alone
 * could pick a stale writer.  It qualifies only if it still writes exactly this matrix and NO op between it and the GroupNorm
 * writes into the buffer (any kind of op: a recycled block rewritten by a LayerNorm / copy / gather would leave the column
 * statistics describing data that is gone). */
static MLOp* gn_producer(MLCtx* C, int gn_idx, int src, const float* x, int64_t ld, int Ci, int64_t rows)
{
	const int j = C->ops[gn_idx].gn_src[src];
	if (j < 0 || j >= gn_idx) return NULL;
	MLOp *o = &C->ops[j];
	if (o->kind != OP_GEMM || o->u.gemm.C32 != x || o->u.gemm.ldc32 != ld || o->u.gemm.N != Ci || o->u.gemm.M != rows ||
	    o->u.gemm.act == MLSD_ACT_GEGLU) return NULL;
	const char *lo = (const char*)x, *hi = lo + (size_t)rows * ld * sizeof(float);
	for (int k=j+1; k<gn_idx; ++k) {
		const void *out[3];
		const int n = op_outputs(&C->ops[k], out);
		for (int q=0;q<n;++q) if ((const char*)out[q] >= lo && (const char*)out[q] < hi) return NULL;
	}
	return o;
}

static int gn_producer_rows(MLOp* o)
{	/* rows per statistics block this producer's kernel would use, 0 if it cannot emit statistics (no side effect) */
	mlsd_gemm_args *g = &o->u.gemm;
	float *keep = g->colstats;
	if (!keep) g->colstats = (float*)(uintptr_t)64;                  /* probe value: alignment is part of the test */
	const int r = mlsd_gemm_colstats_rows(g);
	g->colstats = keep;
	return (r > 0 && !(g->M % r)) ? r : 0;
}

static void wire_gn_stats(MLCtx* C)
{
	const char *e = getenv("MLSD_GN_TWO_PASS");
	if (e && *e && *e != '0') return;
	for (int i=0;i<C->n_ops;++i) {
		if (C->ops[i].kind != OP_GN) continue;
		mlsd_gn_args *g = &C->ops[i].u.gn;
		const int64_t rows = (int64_t)g->n_img * g->HW;
		if (mlsd_groupnorm_single_pass(g->n_img, g->HW, g->C1 + g->C2, g->n_grp)) continue;      /* one dispatch, reads x once: nothing to gain from producer statistics */
		MLOp *p1 = gn_producer(C, i, 0, g->x1, g->ld1, g->C1, rows);
		MLOp *p2 = g->C2 ? gn_producer(C, i, 1, g->x2, g->ld2, g->C2, rows) : NULL;
		if (!p1 || (g->C2 && !p2)) continue;
		const int r1 = gn_producer_rows(p1), r2 = p2 ? gn_producer_rows(p2) : 0;
		if (r1 <= 0 || (g->HW % r1) || (p2 && (r2 <= 0 || (g->HW % r2)))) continue;
		MLOp *pr[2] = { p1, p2 };
		int ok = 1;
		for (int k=0;k<2 && ok;++k) if (pr[k] && !pr[k]->u.gemm.colstats) {     /* one buffer per producer, shared by its consumers */
			mlsd_gemm_args *pg = &pr[k]->u.gemm;
			pg->colstats = (float*)mlctx_dalloc(C, (size_t)(pg->M / (k ? r2 : r1)) * 2 * pg->N * sizeof(float), 1);
			if (!pg->colstats) ok = 0;
		}
		if (!ok) continue;
		g->cs1 = p1->u.gemm.colstats; g->rb_rows1 = r1;
		g->cs2 = p2 ? p2->u.gemm.colstats : NULL; g->rb_rows2 = r2;
		/* the promise the launcher checks at every launch (mlsd_gemm fails if its tile / epilogue no longer writes these blocks) */
		p1->u.gemm.colstats_rows = r1;
		if (p2) p2->u.gemm.colstats_rows = r2;
	}
}

/* LayerNorm at the end of its producer (round 3).  A LayerNorm whose input is the fp32 output of a linear launch that qualifies (mlsd_gemm_ln_fused: single-round
 * 128x320 ping-pong launch, fp32 (+ residual) epilogue) is handed to that launch -- gamma, beta, eps, the fp16 output buffer, scratch for the row blocks' partial
 * statistics and the plan's ticket counters -- and its own op is skipped (MLOp.fused).  Same producer rule as the GroupNorm statistics: the op that DEFINES the input
 * tensor, still writing exactly this matrix, no writer in between.  MLSD_NO_LN_FOLD=1 keeps the separate launches (A/B timing, reference of the parity test). */
static int tt_ln_on(void)
{
	static int on = -1;
	if (on < 0) { const char *e = getenv("MLSD_TT_LN"); on = (e && *e == '0') ? 0 : 1; }
	return on;
}
/* producers moved to the 128 x 160 kernel for a LayerNorm that then could not be handed over (alias rule, scratch) go back to the table's tile */
static void restore_unfolded_promotions(MLCtx* C)
{
	for (int i=0;i<C->n_ops;++i) {
		MLOp *o = &C->ops[i];
		if (o->kind == OP_GEMM && o->saved_variant) { o->u.gemm.tile_variant = o->saved_variant < 0 ? 0 : o->saved_variant; o->saved_variant = 0; }
	}
}

static void wire_ln_fold(MLCtx* C)
{
	const char *e = getenv("MLSD_NO_LN_FOLD");
	if (e && *e && *e != '0') return;
	if (C->cu_budget > 0 && C->cu_budget < 256) return;      /* the tiles of a row block must be resident together: not on a CU-masked stream */
	/* pass 0 sizes the scratch (round 6: a region PER folded launch -- its tiles exchange self-tagged records there, and a tag only has to tell this launch's own
	 * generations apart: 16 bytes per row and column tile, 512 KB for 8192 x 1280), pass 1 hands the LayerNorms over: one region each (the region is what makes a tag unambiguous: it must never be shared between ops) */
	for (int pass=0; pass<2; ++pass) {
		size_t need_max = 0, ws_off = 0;
		int any_splitk = 0, slot = 0;
		for (int i=1;i<C->n_ops;++i) {
			MLOp *l = &C->ops[i];
			if (l->kind != OP_LN || !l->u.ln.y16 || l->u.ln.y32 || !l->u.ln.b) continue;
			const int j = l->gn_src[0];
			if (j != i - 1) continue;                                /* adjacent ops only: the fp16 output is written one op EARLIER now, and the arena may have lent that
			                                                          * block to something that runs in between */
			MLOp *o = &C->ops[j];
			mlsd_gemm_args *g = &o->u.gemm;
			if (o->kind != OP_GEMM || o->once || g->C32 != l->u.ln.x || g->ldc32 != l->u.ln.ldx || g->N != l->u.ln.d || g->M != l->u.ln.rows || g->ln_y16) continue;
			/* two forms (mlsd_gemm_ln_fused): inside a single-round launch of the 128x320 ping-pong tile (needs scratch for the row statistics), or -- round 4 -- in the
			 * reduce pass of a split-K launch (no scratch, no hand-off) */
			/* Round 5: a producer whose tile cannot end with the LayerNorm (the general tiles of SD1.5's 8192 x 320 / 2048 x 640 projections: 30 of its 48 LayerNorms kept a
			 * dispatch of their own) moves to the 128 x 160 two-tiles-per-CU kernel when that kernel takes the shape: its N / 160 partner tiles exchange the row statistics.
			 * Only for N <= 640 (at most 4 partner tiles): in-plan A/B, same box (profiles/r5_gemm_tt_ln_inplan.txt): SD1.5 b1 evaluation -0.8 %, b2 -0.9 % (30 dispatches
			 * fewer); at N = 1280 (8 partners; SDXL b1 / b2's 2048 x 1280 projections) the same move cost +1.0 / +1.2 % and is not made.
			 * MLSD_TT_LN=0 keeps the table's tile and the separate LayerNorm. */
			if (!pass && tt_ln_on() && g->tile_variant != VARIANT_TT && !(g->ksplit > 1) && (!(g->tile_variant == 19 && !(g->N % 320)) && g->N <= 640) &&
			    !(g->M % 128) && !(g->N % 160) && !(g->K & 63) && g->K >= 128 && g->act == MLSD_ACT_NONE && !g->rowbias && !g->bias_m && !g->colstats && !g->C16 &&
			    (!g->conv || (g->KH == 1 && g->KW == 1 && g->stride == 1 && g->pad == 0 && !g->upsample)) && (long)(g->M / 128) * (g->N / 160) <= 512) {
				o->saved_variant = g->tile_variant ? g->tile_variant : -1;
				g->tile_variant = VARIANT_TT;
			}
			const int tt_form = g->tile_variant == VARIANT_TT && !(g->M % 128) && !(g->N % 160) && !(g->ksplit > 1) &&
			                    (long)(g->M / 128) * (g->N / 160) <= 512;      /* (N / 160 partner tiles per row block; at most two blocks per CU: all resident together, gemm_tt.hip) */
			const int pp_form = tt_form || (g->tile_variant != VARIANT_TT && !(g->M % 128) && !(g->N % 320) && !(g->ksplit > 1));
			const size_t need = tt_form ? (size_t)(g->M / 128) * (g->N / 160) * 128 * 16 : pp_form ? (size_t)(g->M / 128) * (g->N / 320) * 128 * 16 : 0;
			if (!pass) { need_max += (need + 255) & ~(size_t)255; if (!pp_form && g->ksplit > 1) any_splitk = 1; continue; }      /* (need_max: the SUM over the candidates) */
			if (ws_off + need > C->ln_ws_bytes || slot >= LN_CNT_WORDS - 1) continue;
			if (!pp_form && !(g->ksplit > 1)) continue;
			{	/* the fp16 rows are written one op EARLIER now, by a launch that is still reading its operands: they must not land in a block the arena lent to one of
				 * them (the attention output `a`, A operand of out_proj, is released right after out_proj is recorded and has exactly the size of the next LayerNorm's
				 * output: ADVICE r3) */
				const char *y0 = (const char*)l->u.ln.y16, *y1 = y0 + (size_t)g->M * g->N * 2;
				const struct { const void* p; size_t n; } rd[4] = {
					{ g->A, (size_t)g->M * (size_t)g->lda * 2 }, { g->resid, g->resid ? (size_t)g->M * (size_t)g->ldr * 4 : 0 },
					{ g->W_, (size_t)g->N * (size_t)g->ldb * 2 }, { g->bias, g->bias ? (size_t)g->N * 4 : 0 } };
				int clash = 0;
				for (int q=0;q<4;++q) if (rd[q].p && rd[q].n && (const char*)rd[q].p < y1 && y0 < (const char*)rd[q].p + rd[q].n) clash = 1;
				/* ONE overlap is sound and common (135 of the 199 folds of the SDXL b4 plan): the rows land EXACTLY on the A operand, row for row (same base, K == N,
				 * same stride).  A tile writes fp16 rows [m0, m0 + 128) only after every tile of that row block has published its statistics, i.e. finished its K
				 * loop, and those tiles are the only readers of A rows [m0, m0 + 128).  Any other overlap is refused. */
				if (clash && !(y0 == (const char*)g->A && g->K == g->N && g->lda == g->N && g->resid != (const float*)g->A &&
				               !((const char*)g->W_ < y1 && y0 < (const char*)g->W_ + rd[2].n))) { C->n_ln_alias++; continue; }
			}
			g->ln_y16 = l->u.ln.y16; g->ldln = g->N; g->ln_gamma = l->u.ln.g; g->ln_beta = l->u.ln.b; g->ln_eps = l->u.ln.eps; g->ln_ws = (float*)((char*)C->ln_ws + ws_off); g->ln_cnt = C->ln_cnt; g->ln_slot = slot;
			if (mlsd_gemm_ln_fused(g) >= 1) {
				l->fused = 1; C->n_ln_fused++; o->folded_from = o->saved_variant; o->saved_variant = 0;
				if (need) { ws_off += (need + 255) & ~(size_t)255; ++slot; }      /* (a LayerNorm in a split-K reduce pass needs neither) */
			}
			else { g->ln_y16 = NULL; g->ln_gamma = g->ln_beta = NULL; g->ln_ws = NULL; g->ln_cnt = NULL; }
		}
		if (!pass) {
			/* (every way out of pass 0 still reaches restore_unfolded_promotions: a producer that pass 0 moved to the 128x160 kernel for a fold that then does not
			 * happen -- nothing to fold, or the counter / workspace allocation failed -- gets the table's tile back) */
			if (!need_max && !any_splitk) break;
			if (!need_max) continue;
			if (!C->ln_cnt) {
				if (mlsd_malloc((void**)&C->ln_cnt, LN_CNT_WORDS * 4)) break;
				if (mlsd_memset(C->ln_cnt, 0, LN_CNT_WORDS * 4, C->stream) || mlsd_stream_sync(C->stream)) break;
			}
			if (C->ln_ws_bytes < need_max) {
				if (C->ln_ws) { mlsd_free(C->ln_ws); C->mem_compute -= C->ln_ws_bytes; }
				C->ln_ws = NULL; C->ln_ws_bytes = 0;
				if (mlsd_malloc((void**)&C->ln_ws, need_max)) break;
				if (mlsd_memset(C->ln_ws, 0, need_max, C->stream) || mlsd_stream_sync(C->stream)) { mlsd_free(C->ln_ws); C->ln_ws = NULL; break; }      /* tags 0: no launch ever writes that one */
				C->ln_ws_bytes = need_max; C->mem_compute += need_max;
			}
		}
	}
	restore_unfolded_promotions(C);
}

static int hoist_on(void);
/* ------------------------------------------------------------------ weight streaming (BASELINE configs[4]; the reference's --unet-split, src/unet.c:390-458)
 * The reference halves the UNet graph and uploads each half's weights before computing it, every evaluation, so that only half the model is resident.  Here
 * (mlctx_set_weight_streaming before the graph is built): weight storage comes from a virtual address range (mlctx_walloc), the master copy sits in pinned host
 * memory in the engine's layout, and at prep the recorded plan is cut into SEGMENTS of consecutive ops whose weights fit one of 3..5 device slabs (n_slab); every weight
 * pointer of an op is replaced by its address inside the slab of its segment.  An evaluation then runs  upload(0), upload(1) | compute(0) | upload(2) into slab 0 ||
 * compute(1) | upload(3) into slab 1 || compute(2) ...: hipMemcpyAsync on a copy stream, ordered against the compute stream by events, so the next segment's weights
 * arrive under the current segment's launches.  Same launches, same operands: results are bit-identical to the resident plan.  Resident instead of streamed: the
 * K/V projection weights of the cross attentions (their launch is step-invariant and hoisted out of the evaluation) and everything that is not a weight. */
static const void** op_weight_slots(MLOp* o, int* n)
{
	static _Thread_local const void** slots[8];
	int k = 0;
	switch (o->kind) {
	case OP_GEMM: slots[k++] = &o->u.gemm.W_; slots[k++] = (const void**)&o->u.gemm.bias; slots[k++] = (const void**)&o->u.gemm.bias_m;
	              slots[k++] = (const void**)&o->u.gemm.ln_gamma; slots[k++] = (const void**)&o->u.gemm.ln_beta;
	              slots[k++] = (const void**)&o->u.gemm.gn_gamma; slots[k++] = (const void**)&o->u.gemm.gn_beta; break;     /* (a GroupNorm folded into the reduce pass: EXPERIMENTS builds) */
	case OP_GN:   slots[k++] = (const void**)&o->u.gn.gamma; slots[k++] = (const void**)&o->u.gn.beta; break;
	case OP_LN:   slots[k++] = (const void**)&o->u.ln.g; slots[k++] = (const void**)&o->u.ln.b; break;
	case OP_CLIP_EMBED: slots[k++] = &o->u.cemb.tw; slots[k++] = (const void**)&o->u.cemb.pw; break;
	default: break;
	}
	*n = k;
	return (const void**)slots;
}

static int pv_find(const MLCtx* C, size_t voff)
{	/* allocation that holds virtual offset voff (allocations are handed out in increasing order) */
	int lo = 0, hi = C->n_pv - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (C->pv_allocs[mid].voff <= voff) lo = mid; else hi = mid - 1; }
	return lo;
}

static void wstream_free(MLCtx* C)
{
	for (int i=0;i<C->n_segs;++i) {
		free(C->segs[i].r);
		if (C->ev_up && C->ev_up[i]) mlsd_event_destroy(C->ev_up[i]);
		if (C->ev_done && C->ev_done[i]) mlsd_event_destroy(C->ev_done[i]);
	}
	free(C->segs); C->segs = NULL; C->n_segs = 0;
	free(C->ev_up); free(C->ev_done); C->ev_up = C->ev_done = NULL;
	if (C->copy_stream) { mlsd_stream_sync(C->copy_stream); mlsd_stream_destroy(C->copy_stream); C->copy_stream = NULL; }   /* (the next evaluation's first segments may be in flight: they read the master and write the slabs freed below) */
	if (C->pmaster) { mlsd_host_free(C->pmaster); C->pmaster = NULL; }
	for (int i=0;i<MLW_NSLAB_MAX;++i) if (C->slab[i]) { mlsd_free(C->slab[i]); C->slab[i] = NULL; }
	C->n_slab = 0;
	C->pf_valid = 0;
	if (C->pscratch) { mlsd_free(C->pscratch); C->pscratch = NULL; C->pscratch_bytes = 0; }
	free(C->pv_allocs); C->pv_allocs = NULL; C->n_pv = C->cap_pv = 0; C->pv_size = 0;
	C->stream_bytes_per_eval = 0;      /* (pstream / slab_bytes are settings of the context, like its flags: they survive mlctx_begin) */
}

/* slab_bytes: size of each of the device slabs (0 = default 512 MiB); call before the graph is built */
MLB_API int mlctx_set_weight_streaming(MLCtx* C, size_t slab_bytes)
{
	if (!C || C->n_ops || C->n_params) return mlctx_fail(C, "mlctx_set_weight_streaming: the graph is already being built");
	C->pstream = 1;
	C->slab_bytes = slab_bytes ? ALIGN_UP(slab_bytes, 256) : ((size_t)512 << 20);
	return 1;
}
MLB_API int mlctx_weight_streaming_copies(const MLCtx* C) { return C && C->pstream ? C->stream_copies_per_eval : 0; }   /* host -> device copies per evaluation (one per segment at best) */
MLB_API int mlctx_weight_streaming_info(const MLCtx* C, int* n_segments, size_t* streamed_bytes_per_eval, size_t* slab_bytes, size_t* host_bytes)
{
	if (!C || !C->pstream) return 0;
	if (n_segments) *n_segments = C->n_segs;
	if (streamed_bytes_per_eval) *streamed_bytes_per_eval = C->stream_bytes_per_eval;
	if (slab_bytes) *slab_bytes = C->slab_bytes;
	if (host_bytes) *host_bytes = C->pv_size;
	return 1;
}

/* at the end of mlctx_prep: segments, slabs, master copy, pointer patch */
static int wstream_setup(MLCtx* C)
{
	if (!C->pstream || !C->pv_size) return 1;
	if (C->flags & MLB_F_HIPGRAPH) return mlctx_fail(C, "weight streaming and hipGraph replay exclude each other");
	/* the largest set of weights one op touches must fit a slab */
	int *seg_of = (int*)malloc(sizeof(int) * (size_t)C->n_ops);
	int *used = (int*)calloc((size_t)C->n_pv, sizeof(int));        /* allocation -> 1 + index in the current segment */
	if (!seg_of || !used) { free(seg_of); free(used); return mlctx_fail(C, "out of memory"); }
	int cap = 16, R = 1;
	C->segs = (MLWSeg*)calloc((size_t)cap, sizeof(MLWSeg)); C->n_segs = 0;
	MLWSeg *cur = NULL;
	for (int i=0;i<C->n_ops && R>0;++i) {
		int ns; const void ***sl = (const void***)op_weight_slots(&C->ops[i], &ns);
		int al[8], na = 0; size_t add = 0;
		for (int q=0;q<ns;++q) {
			const void *pp = *sl[q];
			if (!pp || !is_virtual(C, pp)) continue;
			const int a = pv_find(C, (size_t)((const char*)pp - MLW_VBASE));
			int dup = 0; for (int z=0;z<na;++z) if (al[z] == a) dup = 1;
			if (!dup) { al[na++] = a; if (!cur || !used[a]) add += C->pv_allocs[a].bytes; }
		}
		if (!cur || (na && cur->bytes + add > C->slab_bytes)) {     /* open a new segment at this op */
			if (cur) { cur->op1 = i; for (int z=0;z<cur->n;++z) used[pv_find(C, cur->r[z].voff)] = 0; }
			if (C->n_segs == cap) { cap *= 2; C->segs = (MLWSeg*)realloc(C->segs, sizeof(MLWSeg) * (size_t)cap); memset(C->segs + C->n_segs, 0, sizeof(MLWSeg) * (size_t)(cap - C->n_segs)); }
			cur = &C->segs[C->n_segs++]; cur->op0 = i; cur->n = 0; cur->r = NULL; cur->bytes = 0;
		}
		for (int z=0;z<na;++z) {
			const int a = al[z];
			if (used[a]) continue;
			cur->r = realloc(cur->r, sizeof(*cur->r) * (size_t)(cur->n + 1));
			cur->r[cur->n].voff = C->pv_allocs[a].voff; cur->r[cur->n].bytes = C->pv_allocs[a].bytes; cur->r[cur->n].soff = cur->bytes;
			cur->bytes += C->pv_allocs[a].bytes; used[a] = ++cur->n;
			if (cur->bytes > C->slab_bytes) R = mlctx_fail(C, "weight streaming: op %d needs %zu bytes of weights at once, the slab holds %zu", i, cur->bytes, C->slab_bytes);
		}
		seg_of[i] = C->n_segs - 1;
	}
	if (cur) cur->op1 = C->n_ops;
	/* The host master is laid out in SEGMENT order (an allocation sits where the first segment that needs it expects it), so that a segment's upload is one
	 * contiguous copy -- or a few, where a later segment needs weights again that an earlier one placed: per-tensor copies of 1..50 MB ran at 42.6 GB/s of the
	 * link's 57.6 (tools/h2d_probe.py: 8 MiB chunks 49.5 GB/s, whole-GiB copies 57.6). */
	if (R > 0) {
		size_t run = 0;
		for (int a=0;a<C->n_pv;++a) used[a] = 0;
		for (int g=0; g<C->n_segs; ++g)
			for (int z=0; z<C->segs[g].n; ++z) {
				const int a = pv_find(C, C->segs[g].r[z].voff);
				if (!used[a]) { used[a] = 1; C->pv_allocs[a].moff = run; run += C->pv_allocs[a].bytes; }
				C->segs[g].r[z].moff = C->pv_allocs[a].moff;
			}
		for (int a=0;a<C->n_pv;++a) if (!used[a]) { C->pv_allocs[a].moff = run; run += C->pv_allocs[a].bytes; }      /* (weights no op reads) */
		C->stream_copies_per_eval = 0;
		for (int g=0; g<C->n_segs; ++g)
			for (int z=0; z<C->segs[g].n; ++z)
				if (!z || C->segs[g].r[z].moff != C->segs[g].r[z-1].moff + C->segs[g].r[z-1].bytes || C->segs[g].r[z].soff != C->segs[g].r[z-1].soff + C->segs[g].r[z-1].bytes)
					C->stream_copies_per_eval++;
	}
	if (R > 0) {
		if (mlsd_host_alloc((void**)&C->pmaster, C->pv_size)) R = mlctx_fail(C, "weight streaming: %zu bytes of pinned host memory not available", C->pv_size);
		else memset(C->pmaster, 0, C->pv_size);
	}
	/* How many slabs: the uploads of the NEXT evaluation's first two segments should run under this evaluation's last segment, i.e. the slabs of segments 0 and 1 must not
	 * be the last segment's: (n_segs - 1) % n_slab >= 2.  Smallest count in 3..5 that satisfies it (SDXL, 512 MiB slabs: 9 segments -> 3; with 4 slabs segment 0 would
	 * share its slab with segment 8 and the evaluation period went from 86 to 98 ms). */
	C->n_slab = 3;
	for (int n=3; n<=MLW_NSLAB_MAX; ++n) if (C->n_segs > n && (C->n_segs - 1) % n >= 2) { C->n_slab = n; break; }
	for (int k=0;k<C->n_slab && R>0;++k) {
		if (mlsd_malloc((void**)&C->slab[k], C->slab_bytes)) R = mlctx_fail(C, "weight streaming: slab allocation failed");
		else C->mem_params += C->slab_bytes;
	}
	/* pointer patch: virtual -> slab of the op's segment */
	C->stream_bytes_per_eval = 0;
	for (int g=0; g<C->n_segs && R>0; ++g) {
		MLWSeg *sg = &C->segs[g];
		C->stream_bytes_per_eval += sg->bytes;
		for (int i=sg->op0; i<sg->op1; ++i) {
			int ns; const void ***sl = (const void***)op_weight_slots(&C->ops[i], &ns);
			for (int q=0;q<ns;++q) {
				const void *pp = *sl[q];
				if (!pp || !is_virtual(C, pp)) continue;
				const size_t vo = (size_t)((const char*)pp - MLW_VBASE);
				int z = 0;
				for (; z<sg->n; ++z) if (vo >= sg->r[z].voff && vo < sg->r[z].voff + sg->r[z].bytes) break;
				if (z == sg->n) { R = mlctx_fail(C, "weight streaming: internal (op %d references weights outside its segment)", i); break; }
				*sl[q] = C->slab[g % C->n_slab] + sg->r[z].soff + (vo - sg->r[z].voff);
			}
		}
	}
	/* Nothing may still point into the virtual range: only the slots of op_weight_slots are patched, and mlctx_set_weight_streaming is a public per-context
	 * switch -- a builder whose ops take a parameter through ANOTHER field would launch on a never-mapped address and fault the GPU.  Every pointer-sized word of every
	 * op's argument block is checked (the virtual base is an address no host or device allocation has), and prep fails by name instead. */
	for (int i=0; i<C->n_ops && R>0; ++i) {
		const MLOp *o = &C->ops[i];
		const unsigned char *u = (const unsigned char*)&o->u;
		for (size_t off=0; off + sizeof(void*) <= sizeof(o->u); off += sizeof(void*)) {
			const void *pp; memcpy(&pp, u + off, sizeof(pp));
			if (is_virtual(C, pp)) { R = mlctx_fail(C, "weight streaming: op %d (%s) takes a streamed parameter through a field the streaming plan does not patch "
				"(argument word %zu): this builder cannot run with mlctx_set_weight_streaming", i, o->label, off / sizeof(void*)); break; }
		}
	}
	if (R > 0) {
		C->ev_up = (void**)calloc((size_t)C->n_segs, sizeof(void*)); C->ev_done = (void**)calloc((size_t)C->n_segs, sizeof(void*));
	}
	free(seg_of); free(used);
	return R;
}

/* host address of a streamed parameter's master copy (NULL: the parameter is resident) */
static char* param_master(const MLCtx* C, const MLParam* p)
{
	if (!is_virtual(C, p->dev)) return NULL;
	const size_t vo = (size_t)((const char*)p->dev - MLW_VBASE);
	const MLWAlloc *a = &C->pv_allocs[pv_find(C, vo)];
	return C->pmaster + a->moff + (vo - a->voff);
}

static int compute_streamed(MLCtx* C)
{
	const int ns = C->n_segs;
	if (!C->copy_stream) {
		if (mlsd_stream_create(&C->copy_stream)) return -1;
		for (int g=0; g<ns; ++g) if (mlsd_event_create(&C->ev_up[g]) || mlsd_event_create(&C->ev_done[g])) return -1;
	}
	const int hoist = C->n_once > 0 && hoist_on();
	/* diagnostics (MLSD_WSTREAM_TRACE=1, one plan per process): per segment, when its upload and its ops started and ended on the device, relative to the start of the
	 * evaluation's first op.  Two sets of events, by evaluation parity: the uploads of the NEXT evaluation's first segments are issued (and stamped) in this call. */
	static int trace = -1;
	static void *T0[2][64], *T1[2][64], *TC[2][64];
	if (trace < 0) {
		const char *e = getenv("MLSD_WSTREAM_TRACE"); trace = e && *e && *e != '0';
		if (trace) for (int q=0;q<2;++q) for (int g=0;g<64;++g) if (mlsd_event_create(&T0[q][g]) || mlsd_event_create(&T1[q][g]) || mlsd_event_create(&TC[q][g])) return -1;
	}
	const int tr = trace && ns <= 64, par = C->info.n_compute & 1, NS = C->n_slab;
#define UPLOAD(g, set) do { const MLWSeg *sg_ = &C->segs[g]; \
		if (tr && mlsd_event_record(T0[set][g], C->copy_stream)) return -1; \
		for (int z=0; z<sg_->n; ) { int z1 = z + 1; size_t nb = sg_->r[z].bytes;      /* coalesce ranges contiguous on both sides */ \
			while (z1 < sg_->n && sg_->r[z1].moff == sg_->r[z1-1].moff + sg_->r[z1-1].bytes && sg_->r[z1].soff == sg_->r[z1-1].soff + sg_->r[z1-1].bytes) { nb += sg_->r[z1].bytes; ++z1; } \
			if (mlsd_memcpy(C->slab[(g) % NS] + sg_->r[z].soff, C->pmaster + sg_->r[z].moff, nb, 0, C->copy_stream)) return -1; \
			z = z1; } \
		if (tr && mlsd_event_record(T1[set][g], C->copy_stream)) return -1; \
		if (mlsd_event_record(C->ev_up[g], C->copy_stream)) return -1; } while (0)
	/* Segment g lives in slab g % n_slab and is uploaded as soon as the previous user of that slab (segment g - n_slab, or the last segment of that slab in the
	 * previous evaluation) has finished.  The weights do not change between evaluations, so the first n_slab segments of the NEXT evaluation are uploaded at the end
	 * of this one, behind this evaluation's last users of their slabs: the copy stream never idles across the evaluation boundary (MLSD_WSTREAM_TRACE timeline of the
	 * two-slab form, profiles/r4_wstream_trace.txt: 9 back-to-back uploads of 9.2 ms, then the last segment's 16 ms of ops and the next evaluation's first upload with
	 * nothing beside them: 98 + 7 ms per evaluation for 81 ms of copies).  mlctx_param_set / mlctx_params_synth drop the prefetch (pf_valid). */
#define LAST_USER(k) (ns - 1 - ((ns - 1 - (k)) % NS))      /* last segment of this plan that lives in slab k (k < ns, k < n_slab) */
	const int lead = ns < NS ? ns : NS;
	if (!C->pf_valid)
		for (int g=0; g<lead; ++g) {
			if (C->info.n_compute > 0 && mlsd_stream_wait_event(C->copy_stream, C->ev_done[LAST_USER(g)])) return -1;
			UPLOAD(g, par);
		}
	C->pf_valid = 0;
	for (int g=0; g<ns; ++g) {
		if (mlsd_stream_wait_event(C->stream, C->ev_up[g])) return -1;
		if (tr && mlsd_event_record(TC[par][g], C->stream)) return -1;
		for (int i=C->segs[g].op0; i<C->segs[g].op1; ++i) {
			if (hoist && C->static_valid && C->ops[i].once) continue;
			if (run_op(C, &C->ops[i])) {
				char why[300]; snprintf(why, sizeof(why), "%s", mlsd_last_error());
				mlsd_set_error(-1, "%s: op %d (%s) failed: %s", C->name, i, C->ops[i].label, why);
				return -1;
			}
		}
		if (mlsd_event_record(C->ev_done[g], C->stream)) return -1;
		if (g + NS < ns) {
			if (mlsd_stream_wait_event(C->copy_stream, C->ev_done[g])) return -1;
			UPLOAD(g + NS, par);
		}
	}
	for (int g=0; g<lead; ++g) {                        /* the next evaluation's first segments */
		if (mlsd_stream_wait_event(C->copy_stream, C->ev_done[LAST_USER(g)])) return -1;
		UPLOAD(g, par ^ 1);
	}
	C->pf_valid = 1;
#undef LAST_USER
#undef UPLOAD
	if (tr) {
		mlsd_stream_sync(C->stream);
		fprintf(stderr, "[wstream] evaluation %d, segment: upload start..end | ops start..end (ms from the start of the evaluation's first op; %zu MiB per evaluation)\n", C->info.n_compute, C->stream_bytes_per_eval >> 20);
		for (int g=0; g<ns; ++g) {
			float a = 0, b = 0, c = 0, d = 0;
			mlsd_event_elapsed_ms(TC[par][0], T0[par][g], &a); mlsd_event_elapsed_ms(TC[par][0], T1[par][g], &b);
			mlsd_event_elapsed_ms(TC[par][0], TC[par][g], &c); mlsd_event_elapsed_ms(TC[par][0], C->ev_done[g], &d);
			fprintf(stderr, "[wstream] %2d (%4zu MiB, ops %d..%d): %7.2f .. %7.2f | %7.2f .. %7.2f\n", g, C->segs[g].bytes >> 20, C->segs[g].op0, C->segs[g].op1, a, b, c, d);
		}
	}
	return 1;
}

/* GroupNorm at the end of its producer's split-K reduce pass (round 4; same idea as the LayerNorm form above, no hand-off).  A GroupNorm with ONE source, no raw copy, of
 * a size the one-dispatch GroupNorm takes, whose input is the fp32 output of the op recorded right before it, a split-K launch on the general tiles: the launch gets gamma,
 * beta, eps, groups, rows per image, the SiLU flag and the fp16 output buffer, the GroupNorm op is skipped.  MLSD_NO_GN_FOLD=1 keeps the separate launch (A/B, parity test).
 * MEASURED SLOWER (SD1.5 b1 evaluation 6.84 -> 6.88..6.90 ms): mlsd_gemm_gn_fused answers 0 outside EXPERIMENTS builds, so nothing is folded in the product. */
static void wire_gn_fold(MLCtx* C)
{
	const char *e = getenv("MLSD_NO_GN_FOLD");
	if (e && *e && *e != '0') return;
	const char *eh = getenv("MLSD_GN_FOLD_MAX_HW");       /* A/B knob: largest map (rows per image) whose GroupNorm is folded */
	const int max_hw = eh && *eh ? atoi(eh) : 1024;
	for (int i=1;i<C->n_ops;++i) {
		MLOp *l = &C->ops[i];
		if (l->kind != OP_GN || l->fused) continue;
		mlsd_gn_args *n = &l->u.gn;
		if (n->C2 || n->raw16 || n->cs1 || !n->y16 || l->gn_src[0] != i - 1) continue;
		MLOp *o = &C->ops[i-1];
		mlsd_gemm_args *g = &o->u.gemm;
		if (o->kind != OP_GEMM || o->once || g->C32 != n->x1 || g->ldc32 != n->ld1 || g->N != n->C1 || g->M != n->n_img * n->HW || g->ln_y16 || g->gn_y16 || g->colstats || g->ksplit < 2) continue;
		if (!mlsd_groupnorm_single_pass(n->n_img, n->HW, n->C1, n->n_grp)) continue;
		if (n->HW > max_hw) continue;
		{	/* the fp16 output is written one op earlier, by the reduce pass (the GEMM's own operands are done by then): it must not land on what that pass reads */
			const char *y0 = (const char*)n->y16, *y1 = y0 + (size_t)g->M * g->N * 2;
			const struct { const void* p; size_t nb; } rd[4] = {
				{ g->resid, g->resid ? (size_t)g->M * (size_t)g->ldr * 4 : 0 }, { g->bias, g->bias ? (size_t)g->N * 4 : 0 },
				{ g->rowbias, g->rowbias ? (size_t)(g->M / (g->rows_per_batch > 0 ? g->rows_per_batch : 1)) * (size_t)g->ldrb * 4 : 0 }, { g->ws, g->ws_bytes } };
			int clash = 0;
			for (int q=0;q<4;++q) if (rd[q].p && rd[q].nb && (const char*)rd[q].p < y1 && y0 < (const char*)rd[q].p + rd[q].nb) clash = 1;
			if (clash) continue;
		}
		g->gn_y16 = n->y16; g->gn_ldy = n->C1; g->gn_gamma = n->gamma; g->gn_beta = n->beta; g->gn_eps = n->eps; g->gn_groups = n->n_grp; g->gn_hw = n->HW; g->gn_silu = n->silu;
		if (mlsd_gemm_gn_fused(g) == 1) { l->fused = 1; C->n_gn_fused++; }
		else { g->gn_y16 = NULL; g->gn_gamma = g->gn_beta = NULL; g->gn_groups = g->gn_hw = g->gn_silu = 0; }
	}
}
MLB_API int mlctx_gn_fused(const MLCtx* C) { return C ? C->n_gn_fused : 0; }   /* GroupNorms of the plan that run at the end of their producers' split-K reduce pass */

MLB_API int mlctx_prep(MLCtx* C)
{
	if (C->err) return C->err;
	if (!C->n_names || !C->result) return mlctx_fail(C, "mlctx_prep: empty graph");
	if (!mlt_need32(C, C->result)) return -1;
	if (resolve_names(C) < 0) return -1;
	double fl = 0; unsigned nconv = 0;
	for (int i=0;i<C->n_ops;++i) {
		MLOp *op = &C->ops[i];
		if (op->kind == OP_GEMM) {
			if (!op->u.gemm.C32 && !op->u.gemm.C16 && !op->u.gemm.xa_k) return mlctx_fail(C, "op %d (%s): output never consumed", i, op->label);
			if (op->u.gemm.conv) nconv++;
			/* tile selection: a pure function of the shape (table), unless the offline timing mode is on */
			if (!autotune_on()) { int r = select_gemm(C, op); if (r < 0) return -1; if (r != 1) { C->n_tune_miss++; g_tune_miss++; } }     /* (2: a neighbour's tile; 0: the static rule) */
		}
		fl += op->flops;
	}
	if (C->flags & MLB_F_DUMP) {            /* src/mlblock.c:111-116 */
		char path[96]; snprintf(path, sizeof(path), "dump-graph-%s.txt", C->name[0] ? C->name : "ctx");
		if (mlctx_block_graph_dump_path(C, path) < 0) return -1;
	}
	if (!autotune_on()) { wire_gn_stats(C); wire_ln_fold(C); wire_gn_fold(C); }   /* (need the tiles: the offline tuning mode runs the two-pass / separate forms) */
	C->info.flops = fl; C->info.n_conv = nconv; C->info.n_ops = C->n_ops;
	C->info.mem_params = C->mem_params; C->info.mem_compute = C->mem_compute; C->info.mem_total = C->mem_params + C->mem_compute;
	if (C->err) return C->err;
	if (wstream_setup(C) < 0) return -1;
	C->info.mem_params = C->mem_params; C->info.mem_total = C->mem_params + C->mem_compute;
	C->prepared = 1;
	return 1;
}

/* step-invariant ops (MLOp.once) run once per conditioning; MLSD_NO_HOIST=1 / mlctx_set_hoist(0) runs them every time (A/B, parity test) */
static int g_hoist = -1;
static int hoist_on(void)
{
	if (g_hoist < 0) { const char *e = getenv("MLSD_NO_HOIST"); g_hoist = (e && *e && *e != '0') ? 0 : 1; }
	return g_hoist;
}
MLB_API void mlctx_set_hoist(int on) { g_hoist = on ? 1 : 0; }
MLB_API int mlctx_once_ops(const MLCtx* C) { return C->n_once; }

MLB_API int mlctx_compute(MLCtx* C)
{
	if (!C->prepared) return mlctx_fail(C, "mlctx_compute before mlctx_prep");
	for (int i=0;i<C->n_params;++i) if (!C->params[i].loaded)
		return mlctx_fail(C, "parameter '%s' was never loaded", C->params[i].key);
	double t0 = now_s();
	if (!C->tuned && autotune_on() && !mlsd_runtime_is_dry()) {
		/* (a weight-streaming plan's ops point into slabs that only compute_streamed fills: the eager pass would time -- and return -- garbage, ADVICE r4.  Its GEMM
		 * shapes are those of the resident plan: tune that one.) */
		if (C->pstream && C->n_segs) return mlctx_fail(C, "tile timing mode (MLSD_AUTOTUNE / mlctx_set_autotune) on a weight-streaming context: tune the resident plan, the shapes are the same");
		/* first evaluation: run eagerly, timing the tile variants of every not-yet-seen GEMM shape on its real
		 * operands (the ops before it have already produced them) */
		for (int i=0;i<C->n_ops;++i) {
			if (C->ops[i].kind == OP_GEMM && !C->ops[i].u.gemm.tile_variant && tune_gemm(C, &C->ops[i]) < 0) return -1;
			if (run_op(C, &C->ops[i])) {
				char why[300]; snprintf(why, sizeof(why), "%s", mlsd_last_error());
				mlsd_set_error(-1, "%s: op %d (%s) failed: %s", C->name, i, C->ops[i].label, why);
				return -1;
			}
		}
		C->tuned = 1;
		C->info.n_compute++;
		return 1;
	}
	if (C->pstream && C->n_segs) {
		if (compute_streamed(C) < 0) return -1;
		C->static_valid = 1;
		C->info.t_compute = now_s() - t0;
		C->info.n_compute++;
		return 1;
	}
	int hoist = C->n_once > 0 && hoist_on();
	if (C->flags & MLB_F_HIPGRAPH) {
		if (C->graph_exec) hoist = C->graph_hoisted;     /* what the captured graph leaves out */
		if (hoist && !C->static_valid)          /* the step-invariant ops stay outside the captured graph */
			for (int i=0;i<C->n_ops;++i) if (C->ops[i].once && run_op(C, &C->ops[i])) return mlctx_fail(C, "op %d (%s) failed", i, C->ops[i].label);
		if (!C->graph_exec) {
			if (mlsd_capture_begin(C->stream)) return -1;
			int rc = 0;
			for (int i=0;i<C->n_ops && !rc;++i) if (!(hoist && C->ops[i].once)) rc = run_op(C, &C->ops[i]);
			void *ge = NULL;
			int rc2 = mlsd_capture_end(C->stream, &ge);
			if (rc || rc2) return mlctx_fail(C, "hipGraph capture failed");
			C->graph_exec = ge; C->graph_hoisted = hoist;
		}
		if (mlsd_graph_launch(C->graph_exec, C->stream)) return -1;
	} else {
		for (int i=0;i<C->n_ops;++i) {
			if (hoist && C->static_valid && C->ops[i].once) continue;
			int rc = run_op(C, &C->ops[i]);
			if (rc) {
				char why[300]; snprintf(why, sizeof(why), "%s", mlsd_last_error());
				mlsd_set_error(-1, "%s: op %d (%s) failed: %s", C->name, i, C->ops[i].label, why);
				return -1;
			}
		}
	}
	C->static_valid = 1;
	C->info.t_compute = now_s() - t0;   /* enqueue time: the stream is asynchronous */
	C->info.n_compute++;
	return 1;
}

MLB_API void mlctx_info(const MLCtx* C, MLCtxInfo* out)
{
	*out = C->info;
	out->mem_params = C->mem_params; out->mem_compute = C->mem_compute; out->mem_total = C->mem_params + C->mem_compute;
}

MLB_API int mlctx_op_info(const MLCtx* C, int i, const char** label, double* flops)
{
	if (i < 0 || i >= C->n_ops) return -1;
	if (label) {
		const MLOp *op = &C->ops[i];
		static _Thread_local char buf[96];
		if (op->kind == OP_GEMM) {
			const mlsd_gemm_args *g = &op->u.gemm;
			snprintf(buf, sizeof(buf), "%s%s%s", op->label[0] ? op->label : mlsd_gemm_variant(g), C->flags & MLB_F_OPSHAPES ? " " : "", "");
			if (C->flags & MLB_F_OPSHAPES) {
				size_t l = strlen(buf);
				snprintf(buf + l, sizeof(buf) - l, "%dx%dx%d%s%s", g->M, g->N, g->K, g->C32 ? " f32" : "", g->resid ? "+res" : "");
			}
			*label = buf;
		} else if (op->kind == OP_ATTN && (C->flags & MLB_F_OPSHAPES)) {
			const mlsd_attn_args *a = &op->u.attn;
			snprintf(buf, sizeof(buf), "%s b%d h%d d%d %dx%d", op->label, a->n_batch, a->n_head, a->d_head, a->Tq, a->Tk);
			*label = buf;
		} else if (op->kind == OP_LN && (C->flags & MLB_F_OPSHAPES)) {
			snprintf(buf, sizeof(buf), "%s %dx%d", op->label, op->u.ln.rows, op->u.ln.d);
			*label = buf;
		} else if (op->kind == OP_GN && (C->flags & MLB_F_OPSHAPES)) {
			const mlsd_gn_args *a = &op->u.gn;
			snprintf(buf, sizeof(buf), "%s n%d hw%d c%d+%d%s%s", op->label, a->n_img, a->HW, a->C1, a->C2, a->raw16 ? " +raw" : "",
				a->cs1 ? " stats" : "");   /* "stats": first pass replaced by the producers' column statistics */
			*label = buf;
		} else *label = op->label;
	}
	if (flops) *flops = C->ops[i].flops;
	return 1;
}

/* Algorithmic HBM bytes of op i: every operand read once, every output written once (DESIGN.md "roofline"). */
MLB_API double mlctx_op_bytes(const MLCtx* C, int i)
{
	if (i < 0 || i >= C->n_ops) return 0;
	const MLOp *op = &C->ops[i];
	switch (op->kind) {
	case OP_GEMM: {
		const mlsd_gemm_args *g = &op->u.gemm;
		const double nout = g->act == MLSD_ACT_GEGLU ? g->N / 2 : g->N;
		double b = g->conv ? 2.0 * g->n_img * g->H * g->W * g->Cin : 2.0 * g->M * (double)g->K;
		b += 2.0 * g->N * (double)g->K;
		if (g->bias) b += 4.0 * g->N;
		if (g->rowbias) b += 4.0 * (g->M / (g->rows_per_batch > 0 ? g->rows_per_batch : 1)) * (double)g->N;
		if (g->resid) b += 4.0 * g->M * nout;
		if (g->C32) b += 4.0 * g->M * nout;
		if (g->C16) b += 2.0 * g->M * nout;
		if (g->xa_k) b += 2.0 * g->M * nout + 2.0 * 2.0 * (g->M / g->xa_Tq) * (double)g->xa_Tk * g->N;      /* the attention's output + the images' K and V */
		return b;
	}
	case OP_ATTN: {
		const mlsd_attn_args *a = &op->u.attn;
		const double D = (double)a->n_head * a->d_head;
		return 2.0 * a->n_batch * D * (2.0 * a->Tq + 2.0 * a->Tk);
	}
	case OP_GN: {
		const mlsd_gn_args *a = &op->u.gn;
		if (op->fused) return 0.0;
		const double n = (double)a->n_img * a->HW * (a->C1 + a->C2);
		return n * (4.0 + 2.0 + (a->raw16 ? 2.0 : 0.0));
	}
	case OP_LN: if (op->fused) return 0.0;
		return (double)op->u.ln.rows * op->u.ln.d * (4.0 + (op->u.ln.y16 ? 2.0 : 0.0) + (op->u.ln.y32 ? 4.0 : 0.0));
	case OP_SOFTMAX: return (double)op->u.smax.rows * op->u.smax.cols * 6.0;
	case OP_ACT: return (double)op->u.act.n * 6.0;
	case OP_COPY_F32: return 2.0 * op->u.copy.nbytes;
	default: return 0;
	}
}

MLB_API int mlctx_profile_ops(MLCtx* C, float* ms_out, int n_out)
{
	if (!C->prepared) return mlctx_fail(C, "mlctx_profile_ops before mlctx_prep");
	void *e0 = NULL, *e1 = NULL;
	if (mlsd_event_create(&e0) || mlsd_event_create(&e1)) return -1;
	if (C->pstream && C->copy_stream) { if (mlsd_stream_sync(C->copy_stream)) return -1; }      /* (segments uploaded ahead for the next evaluation: this pass refills the slabs itself) */
	C->pf_valid = 0;
	int seg = 0;
	for (int i=0;i<C->n_ops;++i) {
		if (C->pstream && seg < C->n_segs && C->segs[seg].op0 == i) {     /* streamed weights: this segment's weights into its slab first (blocking: not part of the op's time) */
			const MLWSeg *sg = &C->segs[seg];
			for (int z=0; z<sg->n; ++z)
				if (mlsd_memcpy(C->slab[seg % C->n_slab] + sg->r[z].soff, C->pmaster + sg->r[z].moff, sg->r[z].bytes, 0, C->stream)) return -1;
			if (mlsd_stream_sync(C->stream)) return -1;
			++seg;
		}
		if ((C->ops[i].kind == OP_LN || C->ops[i].kind == OP_GN || C->ops[i].kind == OP_GEMM) && C->ops[i].fused) {     /* runs inside its producer: no launch (an empty event pair would read ~4.8 us) */
			if (i < n_out) ms_out[i] = 0;
			continue;
		}
		mlsd_event_record(e0, C->stream);
		if (run_op(C, &C->ops[i])) return -1;
		mlsd_event_record(e1, C->stream);
		mlsd_event_sync(e1);
		float ms = 0; mlsd_event_elapsed_ms(e0, e1, &ms);
		if (i < n_out) ms_out[i] = ms;
	}
	mlsd_event_destroy(e0); mlsd_event_destroy(e1);
	return C->n_ops;
}

/* ------------------------------------------------------------------ parameters */
MLB_API int mlctx_param_count(const MLCtx* C) { return C->n_params; }
MLB_API int mlctx_params_loaded(const MLCtx* C) { for (int i=0;i<C->n_params;++i) if (!C->params[i].loaded) return 0; return 1; }

MLB_API int mlctx_param_info(const MLCtx* C, int i, const char** key, int* type, int64_t ne[4])
{
	if (i < 0 || i >= C->n_params) return -1;
	if (key) *key = C->params[i].key;
	if (type) *type = C->params[i].type;
	if (ne) memcpy(ne, C->params[i].ne, sizeof(C->params[i].ne));
	return 1;
}

static uint16_t f32_to_f16_rne(float f)
{	/* portable IEEE binary32 -> binary16, round to nearest even (what ggml_fp32_to_fp16_row does) */
	uint32_t x; memcpy(&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u;
	x &= 0x7FFFFFFFu;
	if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((x > 0x7F800000u) ? 0x200u : 0));   /* inf / nan */
	if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                       /* overflow -> inf */
	if (x < 0x33000001u) return (uint16_t)sign;                                                    /* underflow -> 0 */
	int e = (int)(x >> 23) - 127;
	uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
	int shift = e < -14 ? (13 + (-14 - e)) : 13;
	uint32_t hm = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
	if (rem > half || (rem == half && (hm & 1))) hm++;
	uint32_t he = e < -14 ? 0 : (uint32_t)(e + 15);
	uint32_t h = e < -14 ? hm : ((he << 10) + (hm - 0x400u) );
	return (uint16_t)(sign | h);
}

MLB_API uint16_t mlb_f32_to_f16_bits(float f) { return f32_to_f16_rne(f); }

static float f16_to_f32(uint16_t h)
{
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1F, m = h & 0x3FF, x;
	if (e == 0) {
		if (!m) x = sign;
		else { int s = 0; while (!(m & 0x400)) { m <<= 1; s++; } m &= 0x3FF; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
	} else if (e == 31) x = sign | 0x7F800000u | (m << 13);
	else x = sign | ((e + 112) << 23) | (m << 13);
	float f; memcpy(&f, &x, 4); return f;
}

MLB_API float mlb_f16_bits_to_f32(uint16_t h) { return f16_to_f32(h); }

static int64_t param_dst_index(const MLParam* p, int64_t i)
{
	switch (p->layout) {
	case 1: {
		const int64_t k0 = i % p->lp[0]; int64_t t = i / p->lp[0];
		const int64_t k1 = t % p->lp[1]; t /= p->lp[1];
		const int64_t ci = t % p->lp[2], co = t / p->lp[2];
		return ((co * p->lp[1] + k1) * p->lp[0] + k0) * p->lp[4] + ci;
	}
	case 2: {
		const int64_t k = i % p->lp[0], row = i / p->lp[0], d = p->lp[1];
		const int64_t j = row < d ? row : row - d;
		return ((j >> 5) * 64 + (row < d ? 0 : 32) + (j & 31)) * p->lp[0] + k;
	}
	case 3: {
		const int64_t d = p->lp[1], j = i < d ? i : i - d;
		return (j >> 5) * 64 + (i < d ? 0 : 32) + (j & 31);
	}
	default: return i;
	}
}

MLB_API int mlctx_param_set(MLCtx* C, const char* key, int src_type, const void* host, int64_t n_elem)
{
	if (!C->prepared) return mlctx_fail(C, "mlctx_param_set before mlctx_prep (names are resolved at prep)");
	MLParam *p = NULL;
	for (int i=0;i<C->n_params;++i) if (C->params[i].key && !strcmp(C->params[i].key, key)) { p = &C->params[i]; break; }
	if (!p) return mlctx_fail(C, "unknown parameter '%s'", key);
	const int64_t n = p->ne[0]*p->ne[1]*p->ne[2]*p->ne[3];
	if (n != n_elem) return mlctx_fail(C, "parameter '%s': %lld elements given, %lld expected", key, (long long)n_elem, (long long)n);
	const size_t esz = p->type == MLT_F16 ? 2 : 4;
	void *buf = calloc(p->dev_elems, esz);
	if (src_type != MLT_F32 && src_type != MLT_F16 && src_type != MLT_BF16 && src_type != MLT_F64) {
		free(buf);
		return mlctx_fail(C, "parameter '%s': unsupported source type %d", key, src_type);
	}
	for (int64_t i=0;i<n;++i) {
		float v;
		switch (src_type) {
		case MLT_F16: v = f16_to_f32(((const uint16_t*)host)[i]); break;
		case MLT_BF16: { uint32_t u = (uint32_t)((const uint16_t*)host)[i] << 16; memcpy(&v, &u, 4); } break;   /* ggml_bf16_to_fp32_row */
		case MLT_F64: { double d; memcpy(&d, (const char*)host + i*8, 8); v = (float)d; } break;
		default: memcpy(&v, (const char*)host + i*4, 4); break;                                                     /* may be unaligned in an mmap'd file */
		}
		const int64_t o = param_dst_index(p, i);
		if (p->type == MLT_F16) ((uint16_t*)buf)[o] = f32_to_f16_rne(v); else ((float*)buf)[o] = v;
	}
	int rc = 0;
	char *pm = param_master(C, p);
	if (pm) {       /* streamed weight: the master copy is host memory; an evaluation in flight may be uploading from it */
		if (C->copy_stream) { mlsd_stream_sync(C->stream); mlsd_stream_sync(C->copy_stream); }
		C->pf_valid = 0;          /* the next evaluation's first segments were uploaded ahead with the old bytes: upload them again */
		memcpy(pm, buf, p->dev_elems * esz);
	} else {
		rc = mlsd_memcpy(p->dev, buf, p->dev_elems * esz, 0, C->stream);
		if (!rc) rc = mlsd_stream_sync(C->stream);
	}
	free(buf);
	if (rc) return -1;
	p->loaded = 1;
	return 1;
}

/* synthetic weights: restatement of oracle/o_core.c (orc_synth_rule / orc_synth_fill key derivation) */
static uint64_t mix64(uint64_t z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
	z ^= z >> 27; z *= 0x94D049BB133111EBULL;
	z ^= z >> 31; return z;
}

MLB_API int mlctx_params_synth(MLCtx* C, uint64_t seed)
{
	if (!C->prepared) return mlctx_fail(C, "mlctx_params_synth before mlctx_prep");
	double t0 = now_s();
	for (int i=0;i<C->n_params;++i) {
		MLParam *p = &C->params[i];
		int nd = 4; while (nd > 1 && p->ne[nd-1] == 1) nd--;
		float offset = 0, scale;
		const size_t lk = strlen(p->key);
		if (nd == 1) {
			if (lk >= 5 && !strcmp(p->key + lk - 5, ".bias")) { offset = 0; scale = 0.05f; }
			else { offset = 1; scale = 0.1f; }
		} else {
			const double fan_in = nd >= 3 ? (double)p->ne[0]*p->ne[1]*p->ne[2] : (double)p->ne[0];
			scale = (float)(1.0 / sqrt(fan_in));
		}
		uint64_t h = 0xCBF29CE484222325ULL;
		for (const char *s = p->key; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001B3ULL; }
		const uint64_t key = mix64(h ^ (seed * 0x9E3779B97F4A7C15ULL));
		const float kf = (float)((double)scale * 1.7320508075688772 / 65536.0);
		const int64_t n = p->ne[0]*p->ne[1]*p->ne[2]*p->ne[3];
		void *dst = p->dev;
		char *pm = param_master(C, p);
		const size_t pbytes = p->dev_elems * (p->type == MLT_F16 ? 2 : 4);
		if (pm) {       /* streamed weight: fill a device scratch, copy it into the host master (same stream: in order) */
			if (C->pf_valid) { if (C->copy_stream) mlsd_stream_sync(C->copy_stream); C->pf_valid = 0; }      /* (segments uploaded ahead hold the old bytes) */
			if (C->pscratch_bytes < pbytes) {
				if (mlsd_stream_sync(C->stream)) return -1;
				if (C->pscratch) mlsd_free(C->pscratch);
				C->pscratch = NULL; C->pscratch_bytes = 0;
				if (mlsd_malloc(&C->pscratch, ALIGN_UP(pbytes, (size_t)1 << 20))) return -1;
				C->pscratch_bytes = ALIGN_UP(pbytes, (size_t)1 << 20);
			}
			dst = C->pscratch;
			if (p->layout == 1 && p->lp[4] != p->lp[2] && mlsd_memset(dst, 0, pbytes, C->stream)) return -1;
		}
		if (mlsd_synth_fill(dst, p->type == MLT_F16 ? 1 : 0, n, key, offset, kf, p->layout,
				p->lp[0], p->lp[1], p->lp[2], p->lp[3], p->lp[4], C->stream)) return -1;
		if (pm && mlsd_memcpy(pm, dst, pbytes, 1, C->stream)) return -1;
		p->loaded = 1;
	}
	if (mlsd_stream_sync(C->stream)) return -1;
	C->info.t_load = now_s() - t0;
	return 1;
}
