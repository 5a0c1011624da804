/* mlblock_amd.h — the host side (C) of the MI355X engine: a re-creation of the reference's
 * mlblock graph-builder API (src/mlblock.h:82-160, src/mlblock_nn.h:9-41) whose "tensors"
 * are device buffers in channels-last layout and whose "graph" is a recorded launch plan
 * of the HIP kernels in mlsd_kernels.h (replayed per evaluation, optionally as a hipGraph).
 *
 * Same names, argument meaning and error behaviour as the reference where a counterpart
 * exists:  int results, >=1 ok, <0 error (ccommon.h TRY convention), message via
 * mlsd_last_error().  Builder functions return NULL on error and latch the context into an
 * error state that mlctx_prep() reports.
 *
 * Logical shapes follow the reference (ne[0] fastest: activations [W,H,C,N], sequences
 * [d,T,N]); physical storage is [N][H*W][C] (= [N][T][d]) so image<->token reshapes and the
 * head split/merge permutes cost nothing.
 */
#pragma once
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct MLCtx MLCtx;
typedef struct MLTensor MLTensor;

enum { MLT_F32 = 0, MLT_F16 = 1, MLT_Q4_0 = 2, MLT_Q4_1 = 3, MLT_Q5_0 = 6, MLT_Q5_1 = 7, MLT_Q8_0 = 8, MLT_I32 = 26, MLT_I64 = 27, MLT_F64 = 28,
       MLT_BF16 = 30 };  /* ggml_type numbering (mlimgsynth.h:336-339); the Q* block types only occur in GGUF files (dequantised on upload) */

enum MLCtxFlags {           /* src/mlblock.h:29-36 */
	MLB_F_MULTI_COMPUTE = 1,
	MLB_F_QUIET = 2,
	MLB_F_DUMP = 4,
	MLB_F_HIPGRAPH = 8,     /* new: capture the plan into a hipGraph at prep and replay it */
	MLB_F_OPSHAPES = 16,    /* diagnostics: mlctx_op_info labels GEMMs with their MxNxK */
	MLB_F_SYNTH_PARAMS = 32,/* mlctx_run_ fills the parameters with the deterministic synthetic weights (seed 1234) when no tensor store is attached */
};

/* ---- context lifecycle (mlctx_begin/end/prep/compute: src/mlblock.c:54-345) */
MLCtx* mlctx_new(void* stream);           /* stream: hipStream_t or NULL (default stream) */
void   mlctx_destroy(MLCtx* C);
void   mlctx_begin(MLCtx* C, const char* name);
void   mlctx_end(MLCtx* C);
void   mlctx_set_tprefix(MLCtx* C, const char* prefix);   /* C->c.tprefix */
void   mlctx_set_flags(MLCtx* C, int flags);
void   mlctx_set_wtype(MLCtx* C, int wtype);               /* linear weight type of the CHECKPOINT (MLT_F16 | MLT_F32 | MLT_BF16, src/mlimgsynth.c:1242);
                                                              * device storage and MFMA operands are always F16 */
int    mlctx_prep(MLCtx* C);              /* resolve parameter names, finish the plan (result = last tensor) */
int    mlctx_compute(MLCtx* C);           /* replay the plan on the context's stream (asynchronous) */
int    mlctx_sync(MLCtx* C);
int    mlctx_handoff_check(MLCtx* C);  /* 0 / < 0: an in-launch hand-off (stream-K, LayerNorm statistics) of this plan gave up waiting since the last check: results
                                        * invalid; the flag / counter blocks are zeroed again before the error is returned */
/* WEIGHT STREAMING (the reference's --unet-split, src/unet.c:390-458; BASELINE configs[4]): the plan's weights live in pinned host memory and pass through THREE TO FIVE device
 * slabs of slab_bytes each (0 = 512 MiB; the count is chosen at prep from the segment count: 3 for the SDXL UNet's 9 segments), uploaded segment by segment under the previous
 * segments' launches.  Device cost: n_slab x slab_bytes (1.5 - 2.5 GiB at the default) + the resident weights of the step-invariant ops (SDXL: 0.68 GB) -- 2.2 GiB of UNet
 * parameters on the device instead of 4.8; smaller slabs trade VRAM for more, shorter uploads (mlctx_weight_streaming_info reports both).  Call before the graph is built;
 * excludes MLB_F_HIPGRAPH and the tile timing mode.  Only builders whose ops take parameters as GEMM weights / biases / norm affines / embedding tables can stream: prep
 * fails by name for anything else.  Results are bit-identical to the resident plan. */
int    mlctx_set_weight_streaming(MLCtx* C, size_t slab_bytes);
int    mlctx_weight_streaming_copies(const MLCtx* C);   /* host -> device copies per evaluation (the host master is laid out in segment order: one per segment where no weight is needed twice) */
int    mlctx_weight_streaming_info(const MLCtx* C, int* n_segments, size_t* streamed_bytes_per_eval, size_t* slab_bytes, size_t* host_bytes);   /* 0: the plan does not stream */
void   mlctx_set_cus(MLCtx* C, int n);   /* CUs the plan's stream may use (CU-masked stream); < 256: the plan is built without in-launch hand-offs.  Before mlctx_prep */
int    mlctx_ln_alias_refused(const MLCtx* C);
int    mlctx_gn_fused(const MLCtx* C);   /* GroupNorms of the plan that run at the end of their producer's split-K reduce pass (MLSD_NO_GN_FOLD=1: none) */
int    mlctx_handoff_ops(const MLCtx* C);   /* launches of the plan that hand data over inside the launch (0: nothing to check) */
int    mlctx_handoffs_off(MLCtx* C);        /* the plan without in-launch hand-offs (plain tiles, separate LayerNorm launches); returns the ops changed; one-way */
int    mlctx_compute_checked(MLCtx* C);     /* compute + (if the plan has hand-offs) drain, check, and ONE re-run on the hand-off-free plan after a give-up */
int    mlctx_handoff_retries(void);         /* evaluations / generations re-run on the hand-off-free plan in this process */
int    mlctx_debug_raise_giveup(MLCtx* C, int what);   /* test hook: raise the sticky word as a timed-out launch would (0 stream-K, 1 LayerNorm) */
int    mlctx_ln_fused(const MLCtx* C);  /* number of LayerNorms of the plan that run at the END of their producers' launches (mlsd_gemm_args.ln_*; MLSD_NO_LN_FOLD=1: none) */
/* GEMM tile selection is a pure function of the shape (compiled-in table, csrc/host/tune_table.inc), so every process
 * runs the same kernels in the same summation order.  mlctx_set_autotune(1) (or MLSD_AUTOTUNE=1) turns on the OFFLINE
 * timing mode used by tools/tune_all.py to produce that table; mlsd_tune_dump writes the shapes timed in this process. */
void   mlctx_set_autotune(int on);
void   mlctx_set_hoist(int on);             /* 0: run the step-invariant ops (cross-attention K/V of the context) in every evaluation, like the reference's graph */
int    mlctx_once_ops(const MLCtx* C);      /* number of step-invariant ops of the plan */
int    mlctx_plan_tune_misses(const MLCtx* C);   /* the same count for one prepared plan (tests / bench assert 0 on the benchmark plans) */
void   mlctx_set_nearest_tile(int on);          /* 0: exact table hits only (A/B against the static rule); default on, MLSD_NO_NEAREST_TILE=1 switches it off */
int    mlctx_plan_tune_nearest(const MLCtx* C);  /* of those, the shapes a table entry with the NEAREST row count (same N, K, epilogue, geometry; within 4 x) lent its tile to; the rest use the static rule */
int    mlctx_tune_misses(void);           /* GEMM shapes prepared so far that the table does not list (static choice used) */
/* IN-PLAN tuning (offline, tools/tune_inplan.py): times every candidate tile of every GEMM shape of a prepared plan where it runs (op by op, whole
 * plan, `reps` passes per candidate round) and switches a shape when a challenger beats the current choice by 3 %; winners go to the process table
 * (mlsd_tune_dump).  Returns the number of shapes changed, < 0 on error. */
int    mlctx_tune_inplan(MLCtx* C, int reps);
int    mlsd_tune_dump(const char* path);

/* ---- graph definition (src/mlblock.h:115-160) */
void      mlctx_block_begin(MLCtx* C);
MLTensor* mlctx_tensor_add(MLCtx* C, const char* name, MLTensor* t);
MLTensor* mlctx_input_new(MLCtx* C, const char* name, int dtype, int n0, int n1, int n2, int n3);
MLTensor* mlctx_result(MLCtx* C);
/* graph-split marker of the reference (src/mlblock.h:133-139, used by --unet-split): no-op here, returns its argument */
MLTensor* mlctx_split_add(MLCtx* C, MLTensor* t);
void mlctx_free(MLCtx* C);                /* reference name of mlctx_destroy (src/mlblock.h:82) */
int  mlctx_build_alloc(MLCtx* C, MLTensor* result);                 /* step-by-step interface (src/mlblock.h:103): = mlctx_prep for `result` */
int  mlctx_block_graph_dump_path(const MLCtx* C, const char* path); /* the block tree as text (src/mlblock.h:100-101, MLIS_DUMP_GRAPH) */

/* ---- inputs / outputs at the host boundary (ltensor_to/from_backend, src/localtensor.h:96-106).
 * Host data is in the reference layout (ne[0] fastest: NCHW fp32 for images). */
int mlctx_input_set(MLCtx* C, MLTensor* t, const void* host_data, size_t nbytes);
int mlctx_output_get(MLCtx* C, MLTensor* t, float* host_out, size_t nbytes);
/* LocalTensor: the reference's host fp32 tensor (src/localtensor.h:16-27) and the calls the model drivers make on it
 * (ltensor_to_backend / ltensor_from_backend :96-106, ltensor_finite_check src/localtensor.c:63-69, resize / free :46-58,
 * shape check :113-120).  n[0] is the fastest dimension (images: n = {W, H, C, N}, data NCHW; sequences: {C, T, N, 1}). */
typedef struct LocalTensor {
	float* d;
	int n[4];
	int flags;
} LocalTensor;
enum { LT_F_OWNMEM = 1, LT_F_READY = 2 };
size_t ltensor_nelements(const LocalTensor* S);
size_t ltensor_nbytes(const LocalTensor* S);
void   ltensor_resize(LocalTensor* S, int n0, int n1, int n2, int n3);   /* owns (re)allocated memory afterwards */
void   ltensor_free(LocalTensor* S);
int    ltensor_shape_check(const LocalTensor* S, int n0, int n1, int n2, int n3);   /* n# <= 0: any; 1 ok, -1 mismatch */
int    ltensor_finite_check(const LocalTensor* S);                                    /* 1 all finite, -1 otherwise */
int    ltensor_to_backend(MLCtx* C, const LocalTensor* S, MLTensor* input);           /* sizes must agree (the reference asserts) */
int    ltensor_from_backend(MLCtx* C, LocalTensor* S, MLTensor* t);                   /* resizes S to t's reference shape */
/* "all in one" of the reference (src/mlblock.c:324-345): prep, upload the NULL-terminated inputs in the order the plan
 * declared them, compute, read the result (the last tensor) into `out` (may be NULL), then release the plan's memory
 * (mlctx_end).  Parameters must be loadable at prep time: they are taken from the tensor store given to mlctx_tstore_set,
 * or synthesised when MLB_F_SYNTH_PARAMS is set (tests). */
int mlctx_run_(MLCtx* C, LocalTensor* out, const LocalTensor** inputs);
struct MLTStore;
void mlctx_set_tstore(MLCtx* C, const struct MLTStore* S);   /* C->tstore of the reference (src/mlblock.h:58): where mlctx_run_ takes the parameters from */
#define mlctx_run(C,O,...) mlctx_run_((C), (O), (const LocalTensor*[]){ __VA_ARGS__, NULL })
/* device-side access for callers that keep data resident (the sampler): pointer to the input's staging
 * buffer in the reference layout (fp32 NCHW / int32) */
void* mlctx_input_device_ptr(MLTensor* t);
/* image inputs: read the NCHW fp32 source from `dev_src` ([n_src][C][HW]; graph image n uses n % n_src) scaled by
 * dev_scale[n % n_src] (or scale0), instead of the staging buffer.  Must precede the first consumer.
 * (c_in scaling + cond/uncond duplication of src/unet.c:470-472, src/mlimgsynth.c:1578-1582; mode 1 = TAE clamp) */
int mlctx_input_bind(MLTensor* t, const float* dev_src, int n_src, const float* dev_scale, float scale0, int mode);
/* portable fp16 conversion used by the parameter loader (RNE, as ggml_fp32_to_fp16_row) */
uint16_t mlb_f32_to_f16_bits(float f);
float mlb_f16_bits_to_f32(uint16_t h);
const float* mlctx_tensor_device_f32(MLCtx* C, MLTensor* t, int64_t* ld);   /* channels-last fp32 view */
void mlctx_tensor_shape(const MLTensor* t, int64_t ne[4]);

/* ---- parameters (tstore_tensor_read / mlctx_tstore_load: src/mlblock.c:232-292) */
int mlctx_param_count(const MLCtx* C);
int mlctx_params_loaded(const MLCtx* C);       /* 1 when every parameter has been set (synth / param_set / tstore_load) */
/* key = full dotted name as the reference derives it (src/mlblock.c:67-105) */
int mlctx_param_info(const MLCtx* C, int i, const char** key, int* type, int64_t ne[4]);
/* load one parameter from host memory in the REFERENCE layout/shape (element count is what is checked,
 * src/mlblock.c:243); src_type MLT_F32 | MLT_F16 | MLT_BF16 | MLT_F64; converted/repacked to the engine's device layout */
int mlctx_param_set(MLCtx* C, const char* key, int src_type, const void* host_data, int64_t n_elem);
/* deterministic synthetic weights (bench/tests: no checkpoints exist): same generator as oracle/o_core.c */
int mlctx_params_synth(MLCtx* C, uint64_t seed);

/* ---- statistics (MLCtxInfo, src/mlblock.h:74-79) */
typedef struct MLCtxInfo {
	size_t mem_params, mem_compute, mem_total;
	double t_load, t_compute;
	unsigned n_compute, n_conv;
	unsigned n_ops;          /* kernels launched per compute */
	double flops;            /* algorithmic FLOPs per compute (2*MAC of conv, linear, QK^T, PV) */
} MLCtxInfo;
void mlctx_info(const MLCtx* C, MLCtxInfo* out);
/* per-op listing for profiling: returns kernel label, flops of op i */
int mlctx_op_info(const MLCtx* C, int i, const char** label, double* flops);
/* algorithmic HBM bytes of op i (operands read once, outputs written once) */
double mlctx_op_bytes(const MLCtx* C, int i);
/* time every op individually with HIP events (diagnostics; synchronises) */
int mlctx_profile_ops(MLCtx* C, float* ms_out, int n_out);

/* ---- NN blocks (src/mlblock_nn.h:9-41) */
MLTensor* mlb_nn_linear(MLCtx* C, MLTensor* x, int n_out, bool bias);
MLTensor* mlb_nn_conv2d(MLCtx* C, MLTensor* x, int ch_out,
	int k0, int k1, int s0, int s1, int p0, int p1, int d0, int d1, bool bias);
MLTensor* mlb_nn_layer_norm(MLCtx* C, MLTensor* x, bool affine, bool bias, float eps);
MLTensor* mlb_nn_groupnorm(MLCtx* C, MLTensor* x, int n_grp, bool affine, float eps);
static inline MLTensor* mlb_nn_groupnorm32(MLCtx* C, MLTensor* x) { return mlb_nn_groupnorm(C, x, 32, true, 1e-6f); }
MLTensor* mlb_downsample(MLCtx* C, MLTensor* x, int ch_out, bool vae);
MLTensor* mlb_upsample(MLCtx* C, MLTensor* x, int ch_out);
MLTensor* mlb_resnet(MLCtx* C, MLTensor* x, MLTensor* emb, int ch_out);
MLTensor* mlb_GEGLU(MLCtx* C, MLTensor* x, int d_out);
MLTensor* mlb_feed_forward(MLCtx* C, MLTensor* x, int d_out, int mult);
MLTensor* mlb_attn_mhead(MLCtx* C, MLTensor* q, MLTensor* k, MLTensor* v,
	int d_out, int d_embed, int n_head, bool mask, bool bias, bool bias_out);
MLTensor* mlb_basic_transf(MLCtx* C, MLTensor* x, MLTensor* c, int d_out, int d_embed, int n_head);

/* elementwise graph ops the reference builders call on ggml directly */
MLTensor* mlb_silu(MLCtx* C, MLTensor* x);                       /* ggml_silu(_inplace) */
MLTensor* mlb_add(MLCtx* C, MLTensor* a, MLTensor* b);           /* ggml_add, same shape; `a` is consumed */
MLTensor* mlb_concat_ch(MLCtx* C, MLTensor* a, MLTensor* b);     /* ggml_concat(a,b,2), zero-copy */
MLTensor* mlb_timestep_embedding(MLCtx* C, MLTensor* t, int dim, int max_period);
void      mlb_release(MLCtx* C, MLTensor* t);                    /* the caller will not use t again */

#ifdef __cplusplus
}
#endif
