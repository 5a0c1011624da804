/* mlis_abi.h — the public libmlimgsynth C-ABI, as exported by libmlimgsynth_amd.so.
 *
 * Binary-compatible re-declaration of the reference's include/mlimgsynth.h:57-575 (v0.4.x): the enum values, struct layouts
 * and symbol names below ARE the interface its FFI binds (python/mlimgsynth.py:165-208 loads a shared library and calls
 * exactly these), so a program or binding written against the reference header runs on this library unchanged
 * (MLIS_LIB_PATH=.../libmlimgsynth_amd.so).  Each group cites the reference lines it mirrors.  Behavioural differences are
 * listed in INTEGRATION.md section 4 (batches > 1 are supported; the CLIP vocabulary is run-time data found through AUX_DIR;
 * MODEL accepts "synth:<model>" for the synthetic-weight benchmark models).
 */
#ifndef MLIS_ABI_H
#define MLIS_ABI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MLIS_VERSION 0x000402                     /* mlimgsynth.h:60; accepted by mlis_ctx_create_i: 0x000400 <= v < 0x000500 */
#define MLIS_VERSION_STR "0.4.2"

/* mlimgsynth.h:68-77 */
typedef enum { MLIS_E_UNKNOWN = -1, MLIS_E_VERSION = -2, MLIS_E_UNK_OPT = -3, MLIS_E_OPT_VALUE = -4, MLIS_E_PROMPT_PARSE = -5,
               MLIS_E_FILE_NOT_FOUND = -6, MLIS_E_NAN = -7, MLIS_E_IMAGE = -8 } MLIS_ErrorCode;
/* :81-87 */
typedef enum { MLIS_STAGE_IDLE = 0, MLIS_STAGE_COND_ENCODE = 1, MLIS_STAGE_IMAGE_ENCODE = 2, MLIS_STAGE_IMAGE_DECODE = 3,
               MLIS_STAGE_DENOISE = 4 } MLIS_Stage;
/* :91-99, :103-108 */
typedef enum { MLIS_METHOD_NONE = 0, MLIS_METHOD_EULER = 1, MLIS_METHOD_HEUN = 2, MLIS_METHOD_TAYLOR3 = 3, MLIS_METHOD_DPMPP2M = 4,
               MLIS_METHOD_DPMPP2S = 5, MLIS_METHOD__LAST = 5 } MLIS_Method;
typedef enum { MLIS_SCHED_NONE = 0, MLIS_SCHED_UNIFORM = 1, MLIS_SCHED_KARRAS = 2, MLIS_SCHED__LAST = 2 } MLIS_Scheduler;
/* :112-122 */
typedef enum { MLIS_LOGLVL_NONE = 0, MLIS_LOGLVL_ERROR = 10, MLIS_LOGLVL_WARNING = 20, MLIS_LOGLVL_INFO = 30, MLIS_LOGLVL_VERBOSE = 40,
               MLIS_LOGLVL_DEBUG = 50, MLIS_LOGLVL_MAX = 255, MLIS_LOGLVL__INCREASE = 0x100 | 10, MLIS_LOGLVL__DECREASE = 0x200 | 10 } MLIS_LogLvl;
/* :126-138, :142-148 */
typedef enum { MLIS_TENSOR_IMAGE = 1, MLIS_TENSOR_MASK = 2, MLIS_TENSOR_LATENT = 3, MLIS_TENSOR_LMASK = 4, MLIS_TENSOR_COND = 5,
               MLIS_TENSOR_LABEL = 6, MLIS_TENSOR_NCOND = 7, MLIS_TENSOR_NLABEL = 8, MLIS_TENSOR_TMP = 0x100 } MLIS_TensorId;
typedef enum { MLIS_TUF_IMAGE = 1, MLIS_TUF_MASK = 2, MLIS_TUF_LATENT = 4, MLIS_TUF_LMASK = 8, MLIS_TUF_CONDITIONING = 16 } MLIS_TensorUseFlag;
/* :152-158 (+ this library's test models, outside the reference's range) */
typedef enum { MLIS_MODEL_TYPE_NONE = 0, MLIS_MODEL_TYPE_SD1 = 1, MLIS_MODEL_TYPE_SD2 = 2, MLIS_MODEL_TYPE_SDXL = 3, MLIS_MODEL_TYPE__LAST = 3,
               MLIS_MODEL_TYPE_AMD_TINY = 101, MLIS_MODEL_TYPE_AMD_TINYXL = 102, MLIS_MODEL_TYPE_AMD_TINYV = 103 } MLIS_ModelType;
/* :162-169 */
typedef enum { MLIS_SUBMODEL_NONE = 0, MLIS_SUBMODEL_UNET = 1, MLIS_SUBMODEL_VAE = 2, MLIS_SUBMODEL_TAE = 3, MLIS_SUBMODEL_CLIP = 4,
               MLIS_SUBMODEL_CLIP2 = 5 } MLIS_SubModel;
/* :174-346: option ids; argument types as in the reference's comments (str = const char*, float options take double) */
typedef enum {
	MLIS_OPT_NONE = 0, MLIS_OPT_BACKEND = 1, MLIS_OPT_MODEL = 2, MLIS_OPT_TAE = 3, MLIS_OPT_LORA_DIR = 4, MLIS_OPT_LORA = 5,
	MLIS_OPT_LORA_CLEAR = 6, MLIS_OPT_PROMPT = 7, MLIS_OPT_NPROMPT = 8, MLIS_OPT_IMAGE_DIM = 9, MLIS_OPT_BATCH_SIZE = 10,
	MLIS_OPT_CLIP_SKIP = 11, MLIS_OPT_CFG_SCALE = 12, MLIS_OPT_METHOD = 13, MLIS_OPT_SCHEDULER = 14, MLIS_OPT_STEPS = 15,
	MLIS_OPT_F_T_INI = 16, MLIS_OPT_F_T_END = 17, MLIS_OPT_S_NOISE = 18, MLIS_OPT_S_ANCESTRAL = 19, MLIS_OPT_IMAGE = 20,
	MLIS_OPT_IMAGE_MASK = 21, MLIS_OPT_NO_DECODE = 22, MLIS_OPT_TENSOR_USE_FLAGS = 23, MLIS_OPT_SEED = 24, MLIS_OPT_VAE_TILE = 25,
	MLIS_OPT_UNET_SPLIT = 26, MLIS_OPT_THREADS = 27, MLIS_OPT_DUMP_FLAGS = 28, MLIS_OPT_AUX_DIR = 29, MLIS_OPT_CALLBACK = 30,
	MLIS_OPT_ERROR_HANDLER = 31, MLIS_OPT_LOG_LEVEL = 32, MLIS_OPT_MODEL_TYPE = 33, MLIS_OPT_WEIGHT_TYPE = 34,
	MLIS_OPT_NO_PROMPT_PARSE = 35, MLIS_OPT__LAST = 35
} MLIS_Option;

typedef struct MLIS_Ctx MLIS_Ctx;                                                                            /* :352 */
typedef struct MLIS_Image { uint8_t* d; size_t sz; unsigned w, h, c; int flags; } MLIS_Image;                /* :356-363 */
typedef struct MLIS_Progress { MLIS_Stage stage; int step, step_end, nfe; double step_time, time; } MLIS_Progress;   /* :367-374 */
typedef struct MLIS_ErrorInfo { MLIS_ErrorCode code; const char* desc; } MLIS_ErrorInfo;                     /* :378-381 */
typedef struct MLIS_BackendInfo {                                                                            /* :385-394 */
	const char* name; unsigned n_dev;
	struct MLIS_BackendDeviceInfo { const char *name, *desc; size_t mem_free, mem_total; } *devs;
} MLIS_BackendInfo;
typedef struct MLIS_Tensor { float* d; int n[4]; int flags; } MLIS_Tensor;                                   /* :399-403 */
typedef int (*MLIS_Callback)(void*, MLIS_Ctx*, const MLIS_Progress*);                                       /* :408 */
typedef void (*MLIS_ErrorHandler)(void*, MLIS_Ctx*, const MLIS_ErrorInfo*);                                 /* :412 */

#define mlis_ctx_create() mlis_ctx_create_i(MLIS_VERSION)
MLIS_Ctx* mlis_ctx_create_i(int version);                                                                    /* :418-419 */
void mlis_ctx_destroy(MLIS_Ctx** pctx);                                                                      /* :423 */
const char* mlis_errstr_get(const MLIS_Ctx* ctx);                                                            /* :427 */
int mlis_option_set(MLIS_Ctx* ctx, MLIS_Option id, ...);                                                     /* :433 */
int mlis_option_set_str(MLIS_Ctx* ctx, const char* name, const char* value);                                 /* :443 */
int mlis_option_get(MLIS_Ctx* ctx, MLIS_Option id, ...);                                                     /* :450 */
int mlis_generate(MLIS_Ctx* ctx);                                                                            /* :457 */
MLIS_Image* mlis_image_get(MLIS_Ctx* ctx, int idx);                                                          /* :462 */
const char* mlis_infotext_get(MLIS_Ctx* ctx, int idx);                                                       /* :468 */
int mlis_setup(MLIS_Ctx* ctx);                                                                               /* :474 */
MLIS_Tensor* mlis_tensor_get(MLIS_Ctx* ctx, MLIS_TensorId id);                                               /* :479 */
const MLIS_BackendInfo* mlis_backend_info_get(MLIS_Ctx* ctx, unsigned idx, int flags);                       /* :485-486 */
/* :490-508 */
const char* mlis_stage_str(MLIS_Stage id);
const char* mlis_stage_desc(MLIS_Stage id);
MLIS_Stage mlis_stage_fromz(const char* str);
const char* mlis_method_str(MLIS_Method id);
MLIS_Method mlis_method_fromz(const char* str);
const char* mlis_sched_str(MLIS_Scheduler id);
MLIS_Scheduler mlis_sched_fromz(const char* str);
const char* mlis_loglvl_str(MLIS_LogLvl id);
MLIS_LogLvl mlis_loglvl_fromz(const char* str);
const char* mlis_model_type_str(MLIS_ModelType id);
const char* mlis_model_type_desc(MLIS_ModelType id);
MLIS_ModelType mlis_model_type_fromz(const char* str);
const char* mlis_option_str(MLIS_Option id);
MLIS_Option mlis_option_fromz(const char* str);
/* :515-549 */
int mlis_image_encode(MLIS_Ctx* ctx, const MLIS_Tensor* image, MLIS_Tensor* latent, int flags);
int mlis_image_decode(MLIS_Ctx* ctx, const MLIS_Tensor* latent, MLIS_Tensor* image, int flags);
int mlis_mask_encode(MLIS_Ctx* ctx, const MLIS_Tensor* mask, MLIS_Tensor* lmask, int flags);
int mlis_text_tokenize(MLIS_Ctx* ctx, const char* text, int32_t** ptokens, MLIS_SubModel model);
int mlis_clip_text_encode(MLIS_Ctx* ctx, const char* text, MLIS_Tensor* embed, MLIS_Tensor* feat, MLIS_SubModel model, int flags);
enum { MLIS_CTEF_NO_NORM = 1 };
/* :552-558 */
void mlis_tensor_free(MLIS_Tensor*);
size_t mlis_tensor_count(const MLIS_Tensor*);
void mlis_tensor_resize(MLIS_Tensor*, int n0, int n1, int n2, int n3);
void mlis_tensor_resize_like(MLIS_Tensor*, const MLIS_Tensor*);
void mlis_tensor_copy(MLIS_Tensor*, const MLIS_Tensor*);
float mlis_tensor_similarity(const MLIS_Tensor*, const MLIS_Tensor*);

/* ---- additions of this library (not in the reference) */
/* token ids instead of text for the next generation (no vocabulary needed: benchmarks, tests).  Cleared after generation
 * like PROMPT.  weights may be NULL. */
int mlis_amd_prompt_tokens_set(MLIS_Ctx* ctx, const int32_t* tokens, const float* weights, int n, int negative);
/* the engine behind the context (NULL before the first setup/generate): for multi-GPU drivers and profiling */
struct MLIS_AmdCtx* mlis_amd_engine_get(MLIS_Ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
