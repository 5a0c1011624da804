/* Internal structures of the host-side builder/executor (not part of the C-ABI). */
#pragma once
#include "mlblock_amd.h"
#include "mlsd_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MLB_API __attribute__((visibility("default")))

int mlsd_set_error(int code, const char* fmt, ...);

typedef enum {
	OP_GEMM, OP_ATTN, OP_GN, OP_LN, OP_NCHW2NHWC, OP_NHWC2NCHW, OP_TEMB, OP_ACT, OP_CLIP_EMBED, OP_SOFTMAX,
	OP_COPY_F32, OP_XA_VT,
} MLOpKind;

typedef struct MLOp {
	MLOpKind kind;
	double flops;
	char label[56];
	int gn_src[2];          /* OP_GN: index of the op that PRODUCES each fp32 source (MLTensor.prod), -1 = not a GEMM/conv output */
	int fused;              /* OP_LN: the producer's launch ends with this LayerNorm (wire_ln_fold): the op itself does nothing; OP_GEMM: the Linear runs as the second GEMM of its producer's launch */
	int saved_variant;      /* OP_GEMM, during wire_ln_fold: the table's tile of a producer that was moved to the 128 x 160 kernel for its LayerNorm (-1: the static rule); 0 = not moved / fold done */
	int folded_from;        /* OP_GEMM: the tile such a producer had before it was moved AND its LayerNorm was handed over: mlctx_handoffs_off returns it there (-1: the static rule; 0: not moved) */
	int once;               /* step-invariant: depends only on inputs marked static_src (the text conditioning); mlctx_compute
	                         * re-runs it only after such an input was written (mlctx_input_set / mlctx_input_device_ptr) */
	union {
		mlsd_gemm_args gemm;
		mlsd_attn_args attn;
		mlsd_gn_args gn;
		struct { const float* x; int64_t ldx; int rows, d; float eps; const float *g, *b; void* y16; float* y32; } ln;
		struct { const float* src; int n_src, C, HW; void* dst; int n_dst, Cpad; const float* scale; float scale0; int mode; } n2h;
		struct { const float* src; int64_t ld; int n, C, HW; float* dst; float mul, add; } h2n;
		struct { const float* t; int n, dim; float maxp; void* out; } temb;
		struct { const float* x; void* y; size_t n; int act; } act;
		struct { const int32_t* tok; int n, T, d; const void* tw; const float* pw; float* out; } cemb;
		struct { const float* in; int64_t ld_in; void* out; int64_t ld_out; int rows, cols; float scale; } smax;
		struct { const void* src; void* dst; size_t nbytes; } copy;
		struct { const void* v; int64_t ldv; int n_img, Tk, N; void* vt; } xavt;      /* V^T pack of a cross attention that ends its q projection (mlsd_xattn_pack_vt) */
	} u;
} MLOp;

struct MLTensor {
	int64_t ne[4];          /* logical (reference) shape */
	int n, h, w, c;         /* physical: [n][h*w][c] */
	float* d32;  int64_t ld32;
	void*  d16;  int64_t ld16;
	size_t sz32, sz16;      /* arena sizes (0 = not arena owned) */
	int prod;               /* index of the producing GEMM op, or -1 */
	int def_op;             /* ops recorded when the tensor was created: its data is final after op def_op (inputs: -1) */
	MLTensor *cat_a, *cat_b;/* virtual channel concat (both fp32) */
	void* silu16;           /* cached fp16 silu(x) (embedding) */
	size_t sz_silu;
	int is_input, in_type;
	void* in_stage;         /* device staging buffer in reference layout (inputs) */
	size_t in_bytes;
	/* optional override of the NCHW source of an image input (sampler keeps the latent resident) */
	const float* in_src; int in_src_n; const float* in_scale; float in_scale0; int in_mode;
	int released;
	int* dirty;             /* inputs feeding step-invariant ops: points at the owning context's static_valid (cleared on every write) */
	char name[48];
};

typedef struct MLParam {
	char* key;              /* resolved at prep */
	int type;               /* reference type: MLT_F16 / MLT_F32 */
	int64_t ne[4];          /* reference shape (ne[0] fastest) */
	int layout;             /* 0 plain, 1 conv OIHW->OHWI(cin_pad), 2 GEGLU weight interleave, 3 GEGLU bias interleave */
	int64_t lp[5];
	void* dev;              /* device data in engine layout */
	size_t dev_elems;       /* elements allocated (incl. padding) */
	int loaded;
} MLParam;

typedef struct { int kind; char* name; int param; int64_t ne[4]; } MLNameRec;  /* kind: 0 block begin, 1 named op (ne = its shape, 0 if unknown), 2 param */

typedef struct { void* ptr; size_t size; int rel_op; } MLFreeBlk;   /* rel_op: ops recorded at release time */

#define MLW_NSLAB_MAX 5  /* device slabs of a weight-streaming plan: 3..5 (MLCtx.n_slab, chosen from the segment count at prep); segment g lives in slab g % n_slab */
struct MLCtx {
	void* stream;
	char name[64];
	char tprefix[32];
	int flags, wtype;
	int err;
	/* plan */
	MLOp* ops; int n_ops, cap_ops;
	MLTensor** tensors; int n_tensors, cap_tensors;
	MLTensor** inputs; int n_inputs, cap_inputs;
	MLTensor* result;
	MLNameRec* names; int n_names, cap_names;
	MLParam* params; int n_params, cap_params;
	/* device memory */
	void** chunks; int n_chunks, cap_chunks;
	char* cur; size_t cur_left;
	MLFreeBlk* freel; int n_free, cap_free;
	size_t mem_compute, mem_params, mem_peak_live, mem_live;
	void* gn_ws; size_t gn_ws_bytes;
	void* graph_exec;
	/* batched cross-attention K/V projection of the (step-constant) context: one GEMM for all layers */
	struct { MLTensor* ctx; char* wbase; char* out16; int n_in, n_total, n_used; } kvb;
	/* batched time-embedding projections of the resnets (they all read silu(emb)): one GEMM, row-bias slices for the convs */
	struct { MLTensor* emb; char* wbase; float* bbase; float* out32; int n_in, n_total, n_used; } epb;
	const struct MLTStore* tstore;   /* parameters for mlctx_run_ (mlctx_set_tstore) */
	int prepared, tuned, n_tune_miss, n_tune_near;
	int static_valid;       /* the outputs of the `once` ops are current (no static_src input was written since they last ran) */
	int n_once, graph_hoisted;
	int dry;                /* built in the dry runtime: its memory is host memory whatever the mode at destruction */
	void* splitk_ws; size_t splitk_ws_bytes;   /* split-K partial sums / stream-K slabs (one buffer: ops run in order on one stream) */
	unsigned* sk_flags;                        /* stream-K: one flag per persistent block, zeroed once (consumers clear them) */
	unsigned* ln_cnt; float* ln_ws; size_t ln_ws_bytes; int n_ln_fused;   /* LayerNorms ended in their producers (wire_ln_fold): counters, scratch, count */
	int n_gn_fused;            /* GroupNorms ended in their producers' split-K reduce pass (wire_gn_fold) */
	int n_ln_alias, cu_budget;   /* folds refused (output would alias a producer operand); CUs the plan's stream may use (0 = all) */
	/* weight streaming (round 4; BASELINE configs[4], the reference's --unet-split: src/unet.c:390-458).  Weight storage is handed out from a VIRTUAL range, the master copy
	 * lives in pinned host memory, the plan is cut into segments whose weights fit one of n_slab device slabs, and segment i+n_slab is uploaded (copy stream) as soon as segment i is done */
	int pstream;                            /* on (mlctx_set_weight_streaming before the graph is built) */
	size_t pv_size;                         /* bytes of virtual weight space handed out */
	struct MLWAlloc* pv_allocs; int n_pv, cap_pv;
	char* pmaster;                          /* pinned host master copy [pv_size] (engine layout) */
	size_t slab_bytes; char* slab[MLW_NSLAB_MAX]; int n_slab;
	int pf_valid;                           /* the first segments of the NEXT evaluation are already uploaded / in flight (compute_streamed) */
	struct MLWSeg* segs; int n_segs;
	void* copy_stream; void **ev_up, **ev_done;   /* per segment */
	void* pscratch; size_t pscratch_bytes;  /* device scratch for the synthetic fill */
	size_t stream_bytes_per_eval; int stream_copies_per_eval;
	MLCtxInfo info;
};

/* virtual weight addresses: never dereferenced, replaced by slab addresses at prep.  NON-CANONICAL on purpose: wstream_setup scans every 8-byte word of every
 * op's argument block for addresses left in this range, and a packed (int lo, int hi) pair whose high word is a plausible dimension must not look like one -- with the
 * former base 0x6000'0000'0000 the pair (n_img, HW = 24576) of a 192 x 128 latent did, and prep of a streamed SD1.5 / SDXL plan at 1536 x 1024 failed.  The high word is
 * now 0xFFFF6000 = -40960 as an int, a NaN as a float: no dimension, count, stride or scale. */
#define MLW_VBASE ((char*)0xFFFF600000000000ULL)
typedef struct MLWAlloc { size_t voff, bytes, moff; } MLWAlloc;     /* moff: offset in the host master (segment order, wstream_setup) */
typedef struct MLWSeg { int op0, op1; int n; struct { size_t voff, bytes, soff, moff; } *r; size_t bytes; } MLWSeg;
void* mlctx_walloc(MLCtx* C, size_t nbytes);      /* weight storage: device memory, or a virtual address when the plan streams its weights */

/* internal helpers shared by mlblock_nn.c and the model builders */
MLTensor* mlt_new(MLCtx* C, int n, int h, int w, int c);
void*     mlctx_dalloc(MLCtx* C, size_t nbytes, int is_param);
void      mlctx_drelease(MLCtx* C, void* p, size_t nbytes);
MLOp*     mlctx_op_new(MLCtx* C, MLOpKind kind, const char* label);
int       mlctx_fail(MLCtx* C, const char* fmt, ...);
MLParam*  mlctx_param_new(MLCtx* C, const char* name, int type, int64_t n0, int64_t n1, int64_t n2, int64_t n3,
	int layout, int64_t lp0, int64_t lp1);
MLParam*  mlctx_param_new_at(MLCtx* C, const char* name, int type, int64_t n0, int64_t n1, int64_t n2, int64_t n3,
	int layout, void* dev);   /* dev != NULL: parameter lives in caller-provided device memory */
void      mlctx_named_op(MLCtx* C, const char* name);   /* naming record for an op tensor (MLN(name, op)) */
float*    mlt_need32(MLCtx* C, MLTensor* t);
void*     mlt_need16(MLCtx* C, MLTensor* t);

MLTensor* mlctx_input_new_seq(MLCtx* C, const char* name, int dtype, int d, int T, int N);
MLTensor* mlctx_input_new_img(MLCtx* C, const char* name, int w, int h, int c, int n);

/* fused building blocks (mlblock_nn.c) */
typedef struct {
	int act;                 /* MLSD_ACT_* */
	MLTensor* resid;         /* fp32 residual added in the epilogue (same rows x n_out) */
	MLTensor* rowbias;       /* fp32 [n][n_out] added per image (time embedding) */
	int act_post;            /* activation after the residual add (TAESD block) */
} MLEpilogue;
MLTensor* mlb_linear_ex(MLCtx* C, MLTensor* x, int n_out, bool bias, const MLEpilogue* ep, int geglu);
MLTensor* mlb_conv2d_ex(MLCtx* C, MLTensor* x, int ch_out, int k, int s, int p, int upsample, bool bias, const MLEpilogue* ep);
MLTensor* mlb_conv2d_ex2(MLCtx* C, MLTensor* x, int ch_out, int k, int s, int p, int p_end, int upsample, bool bias, const MLEpilogue* ep);
MLTensor* mlb_groupnorm_ex(MLCtx* C, MLTensor* x, int n_grp, float eps, int silu, int want_raw16, MLTensor** raw_out);
MLTensor* mlb_layer_norm_ex(MLCtx* C, MLTensor* x, float eps, int out32);
MLTensor* mlb_attn_mhead_ex(MLCtx* C, MLTensor* q, MLTensor* k, MLTensor* v, int d_out, int d_embed, int n_head,
	bool mask, bool bias, bool bias_out, MLTensor* resid);
MLTensor* mlb_resnet_ex(MLCtx* C, MLTensor* x, MLTensor* emb, int ch_out);
/* announce that `ctx` feeds cross-attention K/V projections totalling n_total output columns: they are computed by
 * ONE GEMM recorded here; mlb_attn_mhead then only takes column slices (parameters keep their per-layer names) */
int mlb_cross_kv_batch(MLCtx* C, MLTensor* ctx, int n_total);
int mlb_emb_proj_batch(MLCtx* C, MLTensor* emb, int n_total);
