/* mlsd_kernels.h — the thin C-ABI shim into the hand-written HIP kernels (gfx950).
 *
 * This is the layer that replaces the ggml op call sites of the reference's hot path
 * (SURVEY.md §2.3).  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * unless stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default).
 * Return convention: 0 = success (like ggml_status / ggml_backend_graph_compute,
 * reference src/mlblock.c:301-307), <0 = error with text in mlsd_last_error().
 *
 * Data layout (MI355X-first, differs from the reference on purpose):
 *   activations  : channels-last.  An image tensor [N,H,W,C] and a token tensor [N,T,C]
 *                  are the same memory, so the reference's permute/cont/reshape nodes
 *                  (src/unet.c:126-137, src/mlblock_nn.c:204-227) disappear.
 *                  Residual streams are fp32; GEMM/conv/attention operands are fp16
 *                  (ggml rounds activations to F16 before every F16-weight mul_mat/im2col,
 *                  SURVEY App. A), accumulation fp32 on MFMA.
 *   weights      : fp16 [n_out][K] row-major, K = n_in (linear) or (kh,kw,cin) with cin
 *                  fastest (conv; repacked from the reference's [cout][cin][kh][kw]).
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- runtime (replaces ggml_backend_*:
 * src/mlimgsynth.c:1131-1161, src/localtensor.h:96-106, src/mlblock.c:257) */
const char* mlsd_last_error(void);
const char* mlsd_peek_runtime_error(void);   /* diagnostics: HIP's pending error text (not cleared), "" if none */
/* dry mode: memory calls are served from host memory so the plan builder can run without a GPU; no kernel
 * can be launched (every launcher fails).  Used by the CPU-only tests of the host logic. */
void mlsd_runtime_dry(int on);
int mlsd_runtime_is_dry(void);
int mlsd_device_count(void);
int mlsd_device_set(int dev);
int mlsd_device_info(int dev, char* name, int name_len, char* arch, int arch_len, int* n_cu,
                     size_t* mem_total, size_t* mem_free);
int mlsd_malloc(void** out, size_t nbytes);
int mlsd_free(void* p);
int mlsd_host_alloc(void** out, size_t nbytes);      /* pinned host memory */
int mlsd_host_free(void* p);
int mlsd_memset(void* dst, int value, size_t nbytes, void* stream);
int mlsd_memcpy(void* dst, const void* src, size_t nbytes, int kind /*0 h2d,1 d2h,2 d2d*/, void* stream);
int mlsd_stream_create(void** out);
int mlsd_stream_create_masked(void** out, const uint32_t* cu_mask, int n_words);   /* kernels of this stream run on the masked CUs only */
int mlsd_cu_census(unsigned* out_dev, int n_blocks, int spin_cycles, void* stream); /* diagnostics: out[block] = (XCC id << 16) | HW_ID[15:0] */
int mlsd_stream_destroy(void* s);
int mlsd_stream_sync(void* s);
int mlsd_device_sync(void);
int mlsd_event_create(void** out);
int mlsd_event_destroy(void* e);
int mlsd_event_record(void* e, void* stream);
int mlsd_event_sync(void* e);
int mlsd_event_elapsed_ms(void* e0, void* e1, float* ms);
int mlsd_stream_wait_event(void* stream, void* e);
int mlsd_capture_begin(void* stream);
int mlsd_capture_end(void* stream, void** graph_exec);
int mlsd_graph_launch(void* graph_exec, void* stream);
int mlsd_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------- RCCL over xGMI (librccl opened lazily with dlopen)
 * The two exchange steps of the image-sharded multi-GPU job: broadcast of the conditioning, gather of the
 * results.  One communicator per process (one process per GPU); the 128-byte unique id from rank 0 reaches the other ranks by
 * any side channel of the launcher.  Buffers are device pointers; calls are asynchronous on `stream`. */
int mlsd_rccl_unique_id(void* out128);
int mlsd_rccl_init(void** comm, int world, int rank, const void* id128);
int mlsd_rccl_destroy(void* comm);
/* the same communicator interface over a HOST transport supplied by the launcher (gloo, MPI, ...): the engine's exchange entry
 * points (mlis_amd_bcast_cond / mlis_amd_gather_results) run unchanged; device buffers are staged through host memory (in the
 * dry runtime they already are host memory).  Callbacks return 0 on success. */
typedef int (*mlsd_host_bcast_fn)(void* user, void* buf, size_t nbytes, int root);
typedef int (*mlsd_host_allgather_fn)(void* user, const void* send, void* recv, size_t nbytes_per_rank);
int mlsd_comm_host(void** comm, int world, int rank, mlsd_host_bcast_fn bcast, mlsd_host_allgather_fn all_gather, void* user);
int mlsd_comm_count(void* comm, int* count, int* kind);   /* ranks as the transport reports them (ncclCommCount); kind: 0 = RCCL, 1 = host transport */
int mlsd_rccl_bcast(void* comm, void* buf, size_t nbytes, int root, void* stream);
int mlsd_rccl_all_gather(void* comm, const void* send, void* recv, size_t nbytes_per_rank, void* stream);

/* ---------------------------------------------------------------- GEMM / implicit-GEMM conv
 * Replaces ggml_mul_mat (+ggml_add bias) at src/mlblock_nn.c:22-25 and ggml_conv_2d (+bias) at
 * src/mlblock_nn.c:44-50, with the adjacent elementwise nodes fused as epilogues:
 *   time-embedding add  src/mlblock_nn.c:141-143     residual add  :154, :242,:247,:251, src/unet.c:143
 *   SiLU src/unet.c:153,159   GELU/quick-GELU src/clip.c:354-356   ReLU src/tae.c:30-37
 *   GEGLU (chunk+gelu+mul) src/mlblock_nn.c:164-169   nearest-2x upsample folded into the gather (:122)
 * C[M,N] = A[M,K] . W[N,K]^T, fp16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16).
 */
enum { MLSD_ACT_NONE = 0, MLSD_ACT_SILU = 1, MLSD_ACT_GELU = 2, MLSD_ACT_GELU_QUICK = 3, MLSD_ACT_RELU = 4,
       MLSD_ACT_GEGLU = 5 };

typedef struct mlsd_gemm_args {
	/* A operand: fp16 activations */
	const void* A;
	int64_t lda;            /* elements between consecutive rows (linear) / pixels (conv) */
	int conv;               /* 0: A is a plain [M][K] matrix; 1: implicit im2col over an NHWC image */
	int n_img, H, W, Cin;   /* conv: source image dims (before the optional 2x upsample) */
	int OH, OW;             /* conv: output dims; M must equal n_img*OH*OW */
	int KH, KW, stride, pad;
	int upsample;           /* conv: 1 = the conv reads nearest-2x-upsampled input (ggml_upscale + conv) */
	/* B operand: fp16 weights [N][ldb], K-contiguous */
	const void* W_;
	int64_t ldb;
	int M, N, K;            /* K % 8 == 0; conv: K = KH*KW*Cin */
	/* epilogue */
	const float* bias;      /* [N] or NULL */
	const float* rowbias;   /* [M/rows_per_batch][ldrb] added per batch element (time embedding) or NULL */
	int rows_per_batch;
	int64_t ldrb;
	const float* resid;     /* fp32 [M][ldr] residual added after activation, or NULL */
	int64_t ldr;
	int act;                /* MLSD_ACT_*; GEGLU: W rows interleaved in blocks of 32 (value,gate), output has N/2 columns */
	float* C32;             /* fp32 output [M][ldc32] or NULL */
	int64_t ldc32;
	void* C16;              /* fp16 output [M][ldc16] or NULL */
	int64_t ldc16;
	const float* bias_m;    /* [M] per-row bias (operands swapped: V^T = Wv . x^T in the VAE attention) or NULL */
	int act_after_resid;    /* 1: activation applied after the residual add (TAESD block: relu(conv + x), src/tae.c:36-37) */
	int tile_variant;       /* 0 = automatic choice; k+1 = use tile variant k (set by the plan's autotuner) */
	/* split-K (small-M, long-K problems that cannot fill 256 CUs with output tiles: the SD1.5 batch-1 convs):
	 * ksplit > 1 slices K over gridDim.y, each slice writes fp32 partial sums to ws, a second kernel sums the
	 * slices in fixed order and applies the epilogue.  Ignored (no split) for GEGLU, when ws is NULL, or when
	 * N / the output strides are not multiples of 4. */
	int ksplit;
	void* ws;               /* >= mlsd_gemm_splitk_ws_bytes(M, N, ksplit) bytes, 16-byte aligned */
	size_t ws_bytes;
	/* column statistics of the fp32 output for a consuming GroupNorm (its first pass disappears): per block of
	 * mlsd_gemm_colstats_rows(args) consecutive rows and per column the sum and the sum of squares of C32, written as
	 * colstats[block][0][n] and colstats[block][1][n] (floats, 2*N per block).  Only the launches for which
	 * mlsd_gemm_colstats_rows() > 0 honour it (ping-pong tiles, fp32 output without activation); NULL = off. */
	float* colstats;
	int colstats_rows;      /* > 0: the consumer was wired for statistics blocks of this many rows: mlsd_gemm FAILS unless this
	                         * launch writes exactly those (a tile / epilogue knob changed between planning and launching) */
	/* 4096 zeroed 32-bit words that stay zero between launches (the kernels that use them clear them again):
	 *  - stream-K (tile variant 19): a flag per persistent block beside the workspace `ws` (>= 256 fp32 slabs of 256 x 256:
	 *    mlsd_gemm_streamk_ws_bytes()).  NULL: the launch runs as variant 17.
	 *  - split-K (ksplit > 1): a ticket counter per output tile; the slices are then added inside the launch (no second launch).
	 *    NULL, or more than 4096 tiles: the two-launch form. */
	unsigned* sk_flags;
	/* LayerNorm at the END of the launch (round 3): when ln_y16 is set and mlsd_gemm_ln_fused(args) says so, the launch also writes
	 * ln_y16[m][n] = fp16(LayerNorm(C32 row m) * ln_gamma + ln_beta) (row stride ldln halfs, eps ln_eps): the LayerNorm that would follow
	 * (ggml_norm + mul + add, src/mlblock_nn.c:65-71) needs no launch of its own.  Linear launches on the 128x320 ping-pong tile (or the 128x160 tile) whose tiles are all
	 * resident at once (M % 128 == 0, N % 320 == 0, (M/128)(N/320) <= 256), fp32 output (+ residual), at most 8 column tiles per row block.  The column tiles of a row
	 * block exchange their row statistics inside the launch as self-tagged records (round 6: no counters, nothing to reset): ln_ws = scratch OF THIS LAUNCH'S OWN
	 * (zeroed once, never shared with another launch: >= M * (N / tile columns) * 16 bytes -- 16 bytes per row and column tile); ln_cnt = a block of 8192 32-bit words
	 * zeroed once and NEVER shared between two ops or shapes (a tile takes the tag of its records from its own record of the op's previous launch); ln_cnt word 8191 = sticky give-up indicator like sk_flags[4095] (ln_slot: unused since the records carry their generation).
	 * mlsd_gemm FAILS when ln_y16 is set and the launch cannot honour it. */
	void* ln_y16; int64_t ldln;
	const float *ln_gamma, *ln_beta;
	float ln_eps;
	float* ln_ws;
	unsigned* ln_cnt;
	int ln_slot;
	/* GroupNorm at the END of a split-K launch's reduce pass (round 4; mlsd_gemm_gn_fused): when gn_y16 is set and the launch qualifies -- general tile, ksplit > 1, fp32
	 * output, N % gn_groups == 0 with N / gn_groups a multiple of 4, an (image, group) slab of gn_hw * N / gn_groups <= 10240 values -- the reduce pass runs one block
	 * per (image, group): slices added in slice order + the epilogue (C32 bit-identical to splitk_reduce), then
	 * gn_y16[m][n] = fp16(silu?((C32 - mean) rstd gn_gamma[n] + gn_beta[n])), statistics over the gn_hw rows of the image and the group's columns
	 * (ggml_group_norm + mul + add [+ silu], src/mlblock_nn.c:86-99,135-136).  M must be a multiple of gn_hw.  mlsd_gemm FAILS when gn_y16 is set and the launch cannot honour it. */
	void* gn_y16; int64_t gn_ldy;
	const float *gn_gamma, *gn_beta;
	float gn_eps;
	int gn_groups, gn_hw, gn_silu;
	/* CROSS ATTENTION at the end of its q projection (round 6; the 128 x 320 ping-pong tile = 128 query rows x 5 heads of 64).  When xa_k is set and
	 * mlsd_gemm_xattn_fused(args) == 1 the launch does NOT store the projection: a tile's q (rounded to fp16 exactly as the unfused launch's C16 would hold it) stays in LDS
	 * and the launch ends with  xa_out[m][64 h + d] = fp16( softmax_key( q_h[m] . K_h[key] / sqrt(64) ) . V_h )  over the xa_Tk <= 80 keys of row m's image -- what
	 * mlb_nn_linear (q_proj) + ggml_nn_attention compute at src/mlblock_nn.c:200-223, src/ggml_extend.c:200-222 for the text context (77 keys).  One dispatch and one
	 * round trip of q through HBM fewer per cross attention (70 per SDXL evaluation).  Operands: xa_k = K [n_img * xa_Tk][xa_ldk] fp16 (key r of image b in row
	 * b * xa_Tk + r; the N columns are the projection's: head h = columns 64 h ..); xa_vt = V TRANSPOSED and zero padded, [n_img][N][96] fp16 (mlsd_xattn_pack_vt, run
	 * once per conditioning); xa_Tq = query rows per image (a multiple of 128: a tile never straddles two images).  Needs N % 320 == 0, M % 128 == 0, fp16 arithmetic
	 * as the unfused pair (fp16 q, K, V, P; fp32 accumulation and softmax).  mlsd_gemm FAILS when xa_k is set and the launch cannot honour it. */
	const void* xa_k; int64_t xa_ldk;
	const void* xa_vt;
	void* xa_out; int64_t xa_ldo;
	int xa_Tq, xa_Tk;
} mlsd_gemm_args;

int mlsd_gemm(const mlsd_gemm_args* a, void* stream);
/* rows per statistics block (64 or 128) if this launch would write a->colstats, 0 if its kernel cannot */
int mlsd_gemm_colstats_rows(const mlsd_gemm_args* a);
/* != 0 if this launch (ln_* fields set) ends with the LayerNorm of its output (see mlsd_gemm_args.ln_y16): 1 = inside the launch, the tiles of a row block exchanging their
 * row statistics (128x320 ping-pong tile; an in-launch hand-off); 2 = in the reduce pass of a split-K launch (one block per finished row: no hand-off; ln_ws / ln_cnt unused) */
int mlsd_gemm_ln_fused(const mlsd_gemm_args* a);
/* 1 if this launch (xa_* fields set) ends with the cross attention of the q it projects: see mlsd_gemm_args.xa_k.  MLSD_XATTN=0 in the environment answers 0 (A/B). */
int mlsd_gemm_xattn_fused(const mlsd_gemm_args* a);
/* 0 = never, 1 = where the fused launch wins (128 x 320 tiles on more than half of the CUs: the default), 2 = wherever the kernel takes the launch (A/B, kernel tests); other: back to MLSD_XATTN */
void mlsd_gemm_set_xattn(int mode);
/* vt[b][n][key] = v[b * Tk + key][n] for key < Tk, 0 for Tk <= key < 96: the V operand of the fused launch (fp16; v row stride ldv halfs; N columns; Tk <= 96) */
int mlsd_xattn_pack_vt(const void* v, int64_t ldv, int n_img, int Tk, int N, void* vt, void* stream);
/* name of the kernel variant mlsd_gemm would pick for these args (for profiling reports) */
/* 1 if this launch (gn_* fields set) ends its split-K reduce pass with the GroupNorm of its output (see mlsd_gemm_args.gn_y16) */
int mlsd_gemm_gn_fused(const mlsd_gemm_args* a);
const char* mlsd_gemm_variant(const mlsd_gemm_args* a);
/* tile order inside an XCD's range: column panels `mode` tiles wide (default 8; 0 = row-major).  A/B timing knob. */
void mlsd_gemm_set_panel(int width);
void mlsd_gemm_set_mode(int mode);     /* older name of mlsd_gemm_set_panel */
/* diagnostics: force a tile variant (-1 = automatic choice) / the scalar epilogue (1) instead of the wide one (0) */
void mlsd_gemm_force_variant(int v);
int mlsd_gemm_num_variants(void);
void mlsd_gemm_set_epilogue(int e);
/* timing-only builds of the main loop (needs -DMLSD_GEMM_EXPERIMENTS; otherwise ignored) */
void mlsd_gemm_set_debug(int d);
/* 1: split-K launches that were given ticket counters (mlsd_gemm_args.sk_flags) add their K slices INSIDE the launch (the block that
 * finishes a tile last sums the slabs in slice order and runs the epilogue: bit-identical to the two-launch form).  Default 0: the
 * two-launch form measured faster on MI355X (profiles/r3_gemm_splitk_inline.txt) */
void mlsd_gemm_set_splitk_inline(int on);
/* split-K launches (ksplit > 1, ticket counters in sk_flags) on the 64x128 / 128x128 tiles add their K slices INSIDE the launch, all blocks of a tile sharing the work
 * (round 4; bit-identical to the two-launch form).  Measured SLOWER than the second launch (a dispatch boundary is the cheaper grid-wide barrier on MI355X): only in
 * builds with EXPERIMENTS=1, and off unless switched on here / by MLSD_SPLITK_PAR=1. */
void mlsd_gemm_set_splitk_parallel(int on);
int mlsd_gemm_splitk_parallel(const mlsd_gemm_args* a);   /* 1 if this launch would do so (an in-launch hand-off: mlctx_handoff_check covers it) */
/* != 0 if mlsd_gemm runs this launch on the small-Cout streaming convolution (tile variant 31, conv_smalln.hip): 3x3, stride 1, pad 1, Cout <= 16, Cin 64 / 128, fp32 output + bias --
 * the KL-VAE decoder's conv_out (src/vae.c:163-165) and TAESD's last layer (src/tae.c:88-89).  MLSD_CONV_SMALLN=0 in the environment keeps the implicit-GEMM tile (A/B). */
int mlsd_conv_smalln_eligible(const mlsd_gemm_args* a);
void mlsd_conv_smalln_set(int ring_rows, int strip_rows);   /* diagnostics: input-row ring depth (4..6) and strip height of the next launches */
size_t mlsd_gemm_streamk_ws_bytes(void);   /* workspace of a stream-K launch (slabs); the flags are 256 x 4 bytes, zeroed once */
void mlsd_gemm_set_cus(int n);      /* CUs a persistent GEMM launch occupies (default 256; 128 for half-chip partitions) */
void mlsd_gemm_set_trace(void* buf);        /* diagnostics: device buffer of 256 x 8 uint64 cycle stamps filled by the ping-pong kernels (NULL = off) */
size_t mlsd_gemm_splitk_ws_bytes(int M, int N, int ksplit);

/* ---------------------------------------------------------------- fused attention
 * Replaces ggml_nn_attention (src/ggml_extend.c:200-222: mul_mat, scale, [diag_mask_inf], soft_max,
 * mul_mat with materialised scores) and the head split/merge permutes around it
 * (src/mlblock_nn.c:204-227).  Flash-style: online softmax, scores never leave the CU.
 * q [n_batch][Tq][ldq], k/v [n_batch][Tk][ldk/ldv] fp16 with head h at column h*d_head; out fp16
 * [n_batch][Tq][ldo], heads merged.  d_head in {40,64,80,160} (any multiple of 8 up to 160). */
typedef struct mlsd_attn_args {
	const void *q, *k, *v;
	void* out;
	int64_t ldq, ldk, ldv, ldo;          /* row strides in elements */
	int64_t bsq, bsk, bsv, bso;          /* batch strides in elements */
	int n_batch, n_head, d_head, Tq, Tk;
	int causal;                           /* 1: key index > query index masked (CLIP, src/clip.c:407-408) */
} mlsd_attn_args;

int mlsd_attention(const mlsd_attn_args* a, void* stream);
/* diagnostics / A-B timing: 1 = the d_head 64 problems also run on the general kernel instead of the 64-rows-per-wave one */
void mlsd_attention_force_old(int on);
void mlsd_attention_x2_min_tq(int tq);     /* smallest Tq (multiple of 256) the 64-rows-per-wave kernel takes (default 2048) */
void mlsd_attention_sp(int mode);          /* d_head 64 or 40, no mask, Tq % 256 == 0, Tk % 64 == 0 and >= 128: 1 (default) = from Tq = 768 on, when the launch has at least 128 blocks of 256 rows, the software-pipelined kernel (attn64x2s_kernel, round 6: Q pre-scaled in fp16), 2 = from Tq = 256 on (kernel tests), 0 = the tile-loop kernels (A/B) */
void mlsd_attention_pp(int mode);          /* d_head 64 ping-pong kernel: 0 off, 1 by shape (default), 2 always 32 rows per wave, 3 always 64, 4 = 32 rows with one block per CU; + 16 / 32: s_setprio 1 around the MFMA clusters / the vector phase (A-B timing) */
void mlsd_attention_tk96(int on, int qb);  /* Tk <= 96 one-pass kernel on/off (A-B timing); qb = 128-row query blocks per workgroup, 0 = automatic */
void mlsd_attention_wide_stores(int on);  /* diagnostics / A-B timing: 0 = the output in 8-byte pieces per lane */
void mlsd_attention_vsum(int on);         /* diagnostics / A-B timing: 1 = row sums of the 64-rows-per-wave kernel on the VALU instead of ones.P MFMAs */

/* row softmax over fp32 scores -> fp16 probabilities (VAE mid attention, d=512 single head,
 * src/vae.c:46-74, computed as GEMM + softmax + GEMM).  in [rows][ld_in] f32, out [rows][ld_out] f16 */
int mlsd_softmax_rows(const float* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols,
                      float scale, void* stream);

/* ---------------------------------------------------------------- normalisation
 * GroupNorm: ggml_group_norm + mul + add (+ silu) at src/mlblock_nn.c:86-99,135-136,146-147.
 * Two fp32 sources (x1 with C1 channels, x2 with C2 channels, either row-strided) give the
 * channel concat of src/unet.c:233 for free.  Output fp16 [n_img][HW][C1+C2]; optional raw fp16 copy
 * of the (concatenated) input for the 1x1 skip conv. `ws` = workspace of mlsd_groupnorm_ws_bytes(). */
typedef struct mlsd_gn_args {
	const float *x1, *x2;
	int64_t ld1, ld2;                     /* pixel strides in elements */
	int C1, C2;                           /* C2 = 0 if no second source */
	int n_img, HW, n_grp;
	float eps;
	const float *gamma, *beta;            /* [C1+C2] */
	int silu;
	void* y16;                            /* fp16 [n_img][HW][C1+C2] */
	void* raw16;                          /* optional fp16 copy of the un-normalised input, or NULL */
	void* ws;
	/* statistics written by the producing GEMMs (mlsd_gemm_args.colstats, [rows / rb_rows][2][C_i]) instead of the first pass
	 * over x: used when every source has them (rb_rows_i > 0) and HW is a multiple of rb_rows_i */
	const float *cs1, *cs2;
	int rb_rows1, rb_rows2;
} mlsd_gn_args;

size_t mlsd_groupnorm_ws_bytes(int n_img, int HW, int n_grp);
int mlsd_groupnorm(const mlsd_gn_args* a, void* stream);
/* 1 if these dimensions run as ONE dispatch (small maps: the (image, group) slab is held in registers; SD1.5 batch 1 is bound by its dispatch count) */
int mlsd_groupnorm_single_pass(int n_img, int HW, int C, int n_grp);
void mlsd_groupnorm_set_single(int on);     /* diagnostics / A-B timing: 0 = always the two-kernel form */
void mlsd_groupnorm_set_finalize2(int on);  /* diagnostics / A-B timing: 0 = the one-level finalize of the producers' statistics also on large maps */

/* LayerNorm over the last dim: ggml_norm + mul + add at src/mlblock_nn.c:65-71.  x fp32 [rows][ldx] ->
 * y fp16 [rows][d] (and/or y32 fp32 [rows][d]) */
int mlsd_layernorm(const float* x, int64_t ldx, int rows, int d, float eps, const float* gamma,
                   const float* beta, void* y16, float* y32, void* stream);
void mlsd_layernorm_stream_blocks(int n);   /* diagnostics / A-B timing: blocks of the streaming form (default 1024), 0 = one row per wave */

/* ---------------------------------------------------------------- small ops */
/* NCHW fp32 [n_src][C][HW] -> NHWC fp16 [n_dst][HW][Cpad] (channels >= C zero-filled);
 * dst image n reads src image n % n_src (cond/uncond duplication of src/mlimgsynth.c:1578-1582);
 * value = f(src * scale[n % n_src]) with scale NULL -> scalar `scale0`
 * (c_in of src/unet.c:470-472; 1/scale_factor of src/vae.c:174);
 * mode 1: 3*tanh(x/3) clamp of src/tae.c:71-73 applied first; mode 2: x*2-1 (sdvae_encoder_pre, src/vae.h:36-40) applied first. */
int mlsd_nchw_to_nhwc_f16(const float* src, int n_src, int C, int HW, void* dst, int n_dst, int Cpad,
                          const float* scale, float scale0, int mode, void* stream);
/* NHWC fp32 [n][HW][ld] (first C channels) -> NCHW fp32 [n][C][HW]; out = in*mul + add
 * ((x+1)/2 of src/vae.h:43-47) */
int mlsd_nhwc_to_nchw_f32(const float* src, int64_t ld, int n, int C, int HW, float* dst, float mul, float add,
                          void* stream);
/* ggml_timestep_embedding(t, dim, 10000) (src/unet.c:150): t fp32 [n] -> fp16 [n][dim] (cos | sin) */
int mlsd_timestep_embedding(const float* t, int n, int dim, float max_period, void* out16, void* stream);
/* y16 = act(x32) elementwise (silu of the embedding, src/mlblock_nn.c:140) */
int mlsd_act_f32_to_f16(const float* x, void* y16, size_t n, int act, void* stream);
/* ggml_get_rows + position add (src/clip.c:337-342): tokens int32 [n][T] -> x fp32 [n][T][d] */
int mlsd_clip_embed(const int32_t* tokens, int n, int T, int d, const void* tok_w16, const float* pos_w,
                    float* out, void* stream);
/* Euler(-ancestral) update with CFG mix on device (src/mlimgsynth.c:1583, src/solvers.c:86,
 * src/sampling.c:170-174):  x[b] += (eps_c*f + eps_u*(1-f)) * dt[b] + noise[b]*s_up[b]
 * x, noise: NCHW fp32 [B][C][HW]; eps: NHWC fp32 [2B][HW][ld] (cond rows 0..B-1, uncond B..2B-1;
 * f<=1: only cond is read).  dt, s_up: fp32 [B] on device. noise may be NULL. */
int mlsd_sampler_update(float* x, const float* eps, int64_t ld, int B, int C, int HW, float cfg,
                        const float* dt, const float* noise, const float* s_up, void* stream);
/* dnsamp_noise_add (src/sampling.c:112-117): x[b] += noise[b] * s[b]; x, noise fp32 [B][per], s fp32 [B] on device */
int mlsd_noise_add(float* x, const float* noise, const float* s, int B, int64_t per, void* stream);
/* ---- general sampler (all solvers / in-painting / img2img / v-prediction): one launch per loop of the reference, same
 * fp32/fp64 operation order (bit-identical to the host arithmetic).  x, dx, tmp vectors: NCHW fp32 [B][C][HW], n = B*C*HW.
 * Scalars by value (they are wave-uniform kernel arguments: nothing to upload or sync). */
/* dx = CFG mix (src/mlimgsynth.c:1565-1587) of the UNet outputs eps NHWC [2B|B][HW][ld]; vparam != 0: each branch first
 * rescaled out*c_out + x_eval*c_skip (src/unet.c:490-494) */
int mlsd_dxdt_cfg(const float* eps, int64_t ld, const float* x_eval, float* dx, int B, int C, int HW, float cfg,
                  int vparam, float c_out, float c_skip, void* stream);
/* fused Euler(-ancestral) step with the CFG mix (one launch per step; the 1-NFE eps-model path):
 * x += (eps_c*f + eps_u*(1-f))*dt [+ noise*s_up]; same operation order as dxdt_cfg + vec_axpy + noise_add_s */
int mlsd_euler_cfg_update(float* x, const float* eps, int64_t ld, int B, int C, int HW, float cfg, float dt,
                          const float* noise, float s_up, void* stream);
int mlsd_vec_axpy(float* out, const float* x, const float* d, float dt, int64_t n, void* stream);     /* out = x + d*dt (solvers.c:86,105,278) */
int mlsd_solver_heun_corr(float* x, const float* dx, const float* d1, float dt, int64_t n, void* stream);   /* solvers.c:112-113 */
int mlsd_solver_taylor3(float* x, const float* dx, float* dp1, float* dp2, float dt, float idtp, float f2, float f3,
                        int64_t n, void* stream);                                                            /* solvers.c:150-165 */
int mlsd_solver_dpmpp2m(float* x, const float* dx, float* dprev, float t_cur, float a, float c, int64_t n, void* stream);   /* :222-229 */
int mlsd_solver_dpmpp2s(float* x, const float* x1, const float* dx1, float t1, float a, int64_t n, void* stream);           /* :281-284 */
int mlsd_fill2_f32(float* a, int na, float va, float* b, int nb, float vb, void* stream);   /* a[0..na) = va, b[0..nb) = vb from kernel arguments (per-evaluation scalars: no copy-engine job) */
int mlsd_noise_add_s(float* x, const float* noise, float s, int64_t n, void* stream);                /* sampling.c:115, scalar sigma */
int mlsd_mask_apply(float* x, const float* x0, const float* mask /*[HW]*/, int HW, int64_t n, void* stream);   /* sampling.c:98-110 */
/* sdvae_latent_sample / sdvae_latent_mean (src/vae.c:188-229): moments NHWC fp32 [B][HW][ld] (mean | logvar) -> latent NCHW
 * [B][cz][HW]; rnd NCHW [B][cz][HW] or NULL (mean only) */
int mlsd_latent_sample(const float* moments, int64_t ld, const float* rnd, float* latent, int B, int cz, int HW, float scale,
                       void* stream);
int mlsd_latent_sample_nchw(const float* moments /* NCHW [B][2cz][HW] */, const float* rnd, float* latent, int B, int cz, int HW,
                            float scale, void* stream);
/* ltensor_copy_slice2 (src/localtensor.h:84-94; VAE tiling, src/vae.c:283-297,368-382) on NCHW fp32 [planes][h][w] tensors */
int mlsd_copy_slice2(float* dst, int dw, int dh, const float* src, int sw, int sh, int n0, int n1, int di0, int di1,
                     int si0, int si1, int planes, void* stream);
/* finite check (ltensor_finite_check, src/unet.c:487): counts non-finite values into *count (device int32) */
int mlsd_count_nonfinite(const float* x, size_t n, int32_t* count, void* stream);
/* deterministic synthetic parameter fill, bit-identical to oracle/o_core.c orc_synth_fill.
 * Writes element i (reference-layout index, ne[0] fastest) at position perm(i):
 *   layout 0: identity;  layout 1: conv weight [k0,k1,cin,cout] (reference) -> [cout][k1][k0][cin_pad];
 *   layout 2: GEGLU row interleave of a [n_in, 2*d] linear weight; layout 3: same for its bias.
 * dtype: 0 fp32, 1 fp16.  The destination must be zero-initialised when padding is present. */
int mlsd_synth_fill(void* dst, int dtype, int64_t n, uint64_t key, float offset, float kf, int layout,
                    int64_t p0, int64_t p1, int64_t p2, int64_t p3, int64_t p4, void* stream);

/* co-issue probe (round 6, tools/coissue_probe.py): 8 x {one 32x32x16 MFMA if mf, nv (3 / 6) vector instructions of kind vk (1 fma, 2 exp, 3 pk_fma, 4 cvt_pk, 5 max3, 6 pk_add, 7 add,
 * 8 dot2_f32_f16, 9 pk_mul, 10 pk_fma_f16, 11 exp_f16)} per iteration; nthreads 256 / 512 = one / two waves per SIMD; clocks[nblocks] */
int mlsd_probe_coissue(const void* src, int iters, int nblocks, int nthreads, int mf, int vk, int nv, void* clocks, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
