/* The public libmlimgsynth API (include/mlis_abi.h == the reference's include/mlimgsynth.h:414-571) on the MI355X engine.
 * Re-creation of the slices of src/mlimgsynth.c that sit on the hot path:
 *   context + options        mlis_ctx_create_i :446-478, mlis_option_set[_str] :787-949 + mlimgsynth_options_set.c.h
 *   setup                    mlis_setup :1250-1299 (backend, model header, identification, weight type)
 *   conditioning             mlis_text_tokenize :1405-1421, mlis_clip_tokens_encode :1423-1468, mlis_text_cond_encode :1501-1563
 *   generation               mlis_generate :1634-1773, mlis_image_get :1775-1793, mlis_infotext_update :1589-1632
 *   codec / tensors          mlis_image_encode/decode :1301-1357, mlis_mask_encode :1359-1365, MLIS_Tensor helpers
 * What differs by design: the graphs, weights and CLIP towers stay RESIDENT between generations (the reference rebuilds
 * and re-uploads them inside every mlis_generate); BATCH_SIZE > 1 works (image i uses seed + i, generate.sh:56-59);
 * the CLIP vocabulary is data found through AUX_DIR (the reference compiles src/clip_merges.c.h in); MODEL may be
 * "synth:<sd1|sd2|sdxl|tiny|tinyxl|tinyv>[:seed]" for the synthetic-weight models used by the benchmark and the tests.
 * LoRA files (kohya naming) are merged into the weights at load time on the host (src/lora.c:9-138).
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include "mlis_abi.h"
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <time.h>
#include <inttypes.h>
#include <sys/stat.h>

#define CTX_SIGNATURE 0x4d4c4953u
enum { CF_USE_TAE = 1, CF_NO_DECODE = 2, CF_NO_PROMPT_PARSE = 4, CF_UNET_SPLIT = 8, CF_WEIGHT_TYPE_SET = 16, CF_MODEL_TYPE_SET = 32 };
enum { READY_BACKEND = 1, READY_MODEL = 2, READY_LORAS = 4 };
enum { LF_PROMPT = 1 };
#define MAX_LORAS 32
#define LT_F_OWNMEM 1      /* src/localtensor.h:22-27 */
#define LT_F_READY 2
#define N_TMP_TENSORS 4
#define MAX_IMAGES 64

struct MLIS_Ctx {
	uint32_t signature;
	/* options (struct c of the reference's MLIS_Ctx, src/mlimgsynth.c:360-395) */
	char *backend, *path_model, *path_tae, *path_aux, *path_lora_dir, *prompt_raw, *nprompt_raw;
	MLISPrompt prompt, nprompt;
	int32_t *ptok[2]; float *ptokw[2]; int n_ptok[2], have_ptok[2];     /* mlis_amd_prompt_tokens_set */
	int model_type, width, height, n_batch, clip_skip, vae_tile, n_thread, dump_flags, flags, tuflags, wtype;
	float cfg_scale;
	int method, sched, n_step;
	float f_t_ini, f_t_end, s_noise, s_ancestral;
	uint64_t seed; uint32_t rng_offset;
	MLIS_Callback callback; void* callback_ud;
	MLIS_ErrorHandler errh; void* errh_ud;
	/* state */
	int rflags;
	struct { char* path; float mult; int flags; } loras[MAX_LORAS]; int n_lora, n_lora_applied;
	char mname[16];
	uint64_t synth_seed; int synth;
	MLTStore *ts, *ts_tae;
	MLIS_AmdCtx* eng; char eng_key[96];
	MLIS_AmdTextCond* tc; char tc_key[64];
	ClipTokenizer* tok;
	MLIS_Tensor image, mask, latent, lmask, cond, label, ncond, nlabel, tmp[N_TMP_TENSORS];
	MLIS_Image imgex[MAX_IMAGES];
	MLIS_BackendInfo backend_info; struct MLIS_BackendDeviceInfo devs[16]; char dev_names[16][2][96];
	MLIS_Progress prg; double t_last;
	char errstr[600];
	char* infotext;
	int32_t* tokens; float* tokens_w; int n_tokens;
	int last_n_step, last_nfe;
};

/* ------------------------------------------------------------------ small helpers */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec*1e-9; }

static void str_set(char** p, const char* s) { free(*p); *p = strdup(s ? s : ""); }
static int str_empty(const char* s) { return !s || !*s; }

static int api_error(MLIS_Ctx* S, int code, const char* fmt, ...)
{	/* ERROR_LOG + mlis_error_handle: text into errstr, the handler (if any) is called */
	va_list ap; va_start(ap, fmt); vsnprintf(S->errstr, sizeof(S->errstr), fmt, ap); va_end(ap);
	if (S->errh) { MLIS_ErrorInfo ei = { (MLIS_ErrorCode)code, S->errstr }; S->errh(S->errh_ud, S, &ei); }
	return code;
}
/* an error produced below the API (text in mlsd_last_error) */
static int api_error_lib(MLIS_Ctx* S, int code) { return api_error(S, code < 0 ? code : MLIS_E_UNKNOWN, "%s", mlsd_last_error()); }

/* identifier comparison of option / enum names: case-insensitive, '-' == '_' (mlimgsynth.h:446-451) */
static int id_eq(const char* a, size_t n, const char* b)
{
	if (strlen(b) != n) return 0;
	for (size_t i=0;i<n;++i) {
		char x = (char)tolower((unsigned char)a[i]), y = b[i];
		if (x == '-') x = '_';
		if (x != y) return 0;
	}
	return 1;
}

/* ------------------------------------------------------------------ enum <-> string tables (src/mlimgsynth.c:210-302) */
static const char* const k_stage[][2] = { {"idle","Idle"}, {"cond_encode","Conditioning encoding"}, {"image_encode","Image encoding"},
	{"image_decode","Image decoding"}, {"denoise","Denoising"} };
static const char* const k_method[] = { "none", "euler", "heun", "taylor3", "dpmpp2m", "dpmpp2s" };
static const char* const k_sched[] = { "none", "uniform", "karras" };
static const char* const k_model[][2] = { {"none","None"}, {"sd1","Stable Diffusion 1.x"}, {"sd2","Stable Diffusion 2.x"}, {"sdxl","Stable Diffusion XL"} };
static const struct { const char* n; int id; } k_loglvl[] = { {"none",0}, {"error",10}, {"warning",20}, {"info",30}, {"verbose",40}, {"debug",50}, {"max",255} };
static const char* const k_option[] = { "none", "backend", "model", "tae", "lora_dir", "lora", "lora_clear", "prompt", "nprompt", "image_dim",
	"batch_size", "clip_skip", "cfg_scale", "method", "scheduler", "steps", "f_t_ini", "f_t_end", "s_noise", "s_ancestral", "image",
	"image_mask", "no_decode", "tensor_use_flags", "seed", "vae_tile", "unet_split", "threads", "dump_flags", "aux_dir", "callback",
	"error_handler", "log_level", "model_type", "weight_type", "no_prompt_parse" };
#define COUNTOF(a) ((int)(sizeof(a)/sizeof((a)[0])))

static int from_list(const char* const* list, int n, int stride, const char* s, size_t len)
{
	for (int i=0;i<n;++i) if (id_eq(s, len, list[i*stride])) return i;
	return -1;
}
MLB_API const char* mlis_stage_str(MLIS_Stage x) { return (x >= 0 && x < COUNTOF(k_stage)) ? k_stage[x][0] : "???"; }
MLB_API const char* mlis_stage_desc(MLIS_Stage x) { return (x >= 0 && x < COUNTOF(k_stage)) ? k_stage[x][1] : "???"; }
MLB_API MLIS_Stage mlis_stage_fromz(const char* s) { return (MLIS_Stage)from_list(&k_stage[0][0], COUNTOF(k_stage), 2, s, strlen(s)); }
MLB_API const char* mlis_method_str(MLIS_Method x) { return (x >= 0 && x < COUNTOF(k_method)) ? k_method[x] : "???"; }
MLB_API MLIS_Method mlis_method_fromz(const char* s) { return (MLIS_Method)from_list(k_method, COUNTOF(k_method), 1, s, strlen(s)); }
MLB_API const char* mlis_sched_str(MLIS_Scheduler x) { return (x >= 0 && x < COUNTOF(k_sched)) ? k_sched[x] : "???"; }
MLB_API MLIS_Scheduler mlis_sched_fromz(const char* s) { return (MLIS_Scheduler)from_list(k_sched, COUNTOF(k_sched), 1, s, strlen(s)); }
MLB_API const char* mlis_loglvl_str(MLIS_LogLvl id) { for (int i=0;i<COUNTOF(k_loglvl);++i) if (k_loglvl[i].id == (int)id) return k_loglvl[i].n; return "???"; }
MLB_API MLIS_LogLvl mlis_loglvl_fromz(const char* s) { for (int i=0;i<COUNTOF(k_loglvl);++i) if (id_eq(s, strlen(s), k_loglvl[i].n)) return (MLIS_LogLvl)k_loglvl[i].id; return (MLIS_LogLvl)-1; }
MLB_API const char* mlis_option_str(MLIS_Option x) { return (x >= 0 && x < COUNTOF(k_option)) ? k_option[x] : "???"; }
MLB_API MLIS_Option mlis_option_fromz(const char* s) { return (MLIS_Option)from_list(k_option, COUNTOF(k_option), 1, s, strlen(s)); }

static const char* model_name(int mt)
{
	switch (mt) { case MLIS_MODEL_TYPE_SD1: return "sd1"; case MLIS_MODEL_TYPE_SD2: return "sd2"; case MLIS_MODEL_TYPE_SDXL: return "sdxl";
	case MLIS_MODEL_TYPE_AMD_TINY: return "tiny"; case MLIS_MODEL_TYPE_AMD_TINYXL: return "tinyxl"; case MLIS_MODEL_TYPE_AMD_TINYV: return "tinyv"; }
	return NULL;
}
static int model_from(const char* s, size_t n)
{
	int r = from_list(&k_model[0][0], COUNTOF(k_model), 2, s, n);
	if (r >= 0) return r;
	if (id_eq(s, n, "tiny")) return MLIS_MODEL_TYPE_AMD_TINY;
	if (id_eq(s, n, "tinyxl")) return MLIS_MODEL_TYPE_AMD_TINYXL;
	if (id_eq(s, n, "tinyv")) return MLIS_MODEL_TYPE_AMD_TINYV;
	return -1;
}
MLB_API const char* mlis_model_type_str(MLIS_ModelType x) { const char *m = model_name(x); return m ? m : (x == 0 ? "none" : "???"); }
MLB_API const char* mlis_model_type_desc(MLIS_ModelType x) { return (x >= 0 && x < COUNTOF(k_model)) ? k_model[x][1] : (model_name(x) ? "test model" : "???"); }
MLB_API MLIS_ModelType mlis_model_type_fromz(const char* s) { return (MLIS_ModelType)model_from(s, strlen(s)); }

/* ------------------------------------------------------------------ MLIS_Tensor helpers (src/localtensor.h) */
MLB_API size_t mlis_tensor_count(const MLIS_Tensor* t) { return t ? (size_t)t->n[0]*t->n[1]*t->n[2]*t->n[3] : 0; }
MLB_API void mlis_tensor_free(MLIS_Tensor* t) { if (t) { if (t->flags & LT_F_OWNMEM) free(t->d); memset(t, 0, sizeof(*t)); } }
MLB_API void mlis_tensor_resize(MLIS_Tensor* t, int n0, int n1, int n2, int n3)
{
	const size_t n = (size_t)n0*n1*n2*n3;
	if (!(t->flags & LT_F_OWNMEM)) t->d = NULL;
	t->d = (float*)realloc(t->d, (n ? n : 1) * sizeof(float));
	t->n[0]=n0; t->n[1]=n1; t->n[2]=n2; t->n[3]=n3;
	t->flags |= LT_F_OWNMEM;
}
MLB_API void mlis_tensor_resize_like(MLIS_Tensor* t, const MLIS_Tensor* s) { mlis_tensor_resize(t, s->n[0], s->n[1], s->n[2], s->n[3]); }
MLB_API void mlis_tensor_copy(MLIS_Tensor* d, const MLIS_Tensor* s) { mlis_tensor_resize_like(d, s); memcpy(d->d, s->d, mlis_tensor_count(s)*sizeof(float)); }
MLB_API float mlis_tensor_similarity(const MLIS_Tensor* a, const MLIS_Tensor* b)
{	/* cosine similarity (ltensor_similarity) */
	const size_t n = mlis_tensor_count(a);
	if (n != mlis_tensor_count(b) || !n) return 0;
	double ab = 0, aa = 0, bb = 0;
	for (size_t i=0;i<n;++i) { ab += (double)a->d[i]*b->d[i]; aa += (double)a->d[i]*a->d[i]; bb += (double)b->d[i]*b->d[i]; }
	return (float)(ab / sqrt(aa * bb));
}
static int tensor_good(const MLIS_Tensor* t) { return t->d && mlis_tensor_count(t) > 0; }

static int lora_add(MLIS_Ctx* S, const char* name, size_t len, float mult, int flags);
static void loras_remove(MLIS_Ctx* S, int only_prompt);

/* ------------------------------------------------------------------ context */
MLB_API MLIS_Ctx* mlis_ctx_create_i(int version)
{
	if (!(0x000400 <= version && version < 0x000500)) { mlsd_set_error(MLIS_E_VERSION, "mlis incompatible version %06x", version); return NULL; }
	unet_params_init();
	MLIS_Ctx *S = (MLIS_Ctx*)calloc(1, sizeof(*S));
	if (!S) return NULL;
	S->signature = CTX_SIGNATURE;
	S->wtype = MLT_F16;
	S->cfg_scale = 7;
	S->f_t_ini = 1;
	struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
	S->seed = (uint64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;      /* g_rng.seed = timing_timeofday()*1000 (:458) */
	return S;
}

static void engine_drop(MLIS_Ctx* S)
{
	/* (the context's Philox offset is copied back after every SEEDED use of an engine -- generate / image encode -- and nowhere else:
	 * an engine that was only built for mlis_image_decode still has offset 0 and must not reset the context's running offset) */
	if (S->eng) { mlis_amd_destroy(S->eng); S->eng = NULL; S->eng_key[0] = 0; }
}
static void textcond_drop(MLIS_Ctx* S) { if (S->tc) { mlis_amd_textcond_destroy(S->tc); S->tc = NULL; S->tc_key[0] = 0; } }
static void model_drop(MLIS_Ctx* S)
{
	engine_drop(S); textcond_drop(S);
	if (S->ts) { mlts_close(S->ts); S->ts = NULL; }
	if (S->ts_tae) { mlts_close(S->ts_tae); S->ts_tae = NULL; }
	S->rflags &= ~READY_MODEL;
}

MLB_API void mlis_ctx_destroy(MLIS_Ctx** pctx)
{
	if (!pctx || !*pctx) return;
	MLIS_Ctx *S = *pctx;
	if (S->signature != CTX_SIGNATURE) return;
	model_drop(S);
	if (S->tok) clip_tokr_free(S->tok);
	free(S->backend); free(S->path_model); free(S->path_tae); free(S->path_aux); free(S->path_lora_dir); free(S->prompt_raw); free(S->nprompt_raw);
	mlis_prompt_free(&S->prompt); mlis_prompt_free(&S->nprompt);
	for (int i=0;i<2;++i) { free(S->ptok[i]); free(S->ptokw[i]); }
	MLIS_Tensor *ts[] = { &S->image, &S->mask, &S->latent, &S->lmask, &S->cond, &S->label, &S->ncond, &S->nlabel };
	for (int i=0;i<8;++i) mlis_tensor_free(ts[i]);
	for (int i=0;i<N_TMP_TENSORS;++i) mlis_tensor_free(&S->tmp[i]);
	for (int i=0;i<MAX_IMAGES;++i) if (S->imgex[i].flags & LT_F_OWNMEM) free(S->imgex[i].d);
	free(S->infotext); free(S->tokens); free(S->tokens_w);
	loras_remove(S, 0);
	S->signature = 0;
	free(S);
	*pctx = NULL;
}

MLB_API const char* mlis_errstr_get(const MLIS_Ctx* S) { return S ? S->errstr : mlsd_last_error(); }
MLB_API struct MLIS_AmdCtx* mlis_amd_engine_get(MLIS_Ctx* S) { return S ? S->eng : NULL; }

/* ------------------------------------------------------------------ options */
/* by_option: the type comes from MLIS_OPT_MODEL_TYPE (it then survives a checkpoint whose type cannot be detected); a type that
 * mlis_setup DETECTED describes that checkpoint only and must not excuse the next, unidentifiable one */
static int model_type_set_ex(MLIS_Ctx* S, int mt, int by_option);
static int model_type_set(MLIS_Ctx* S, int mt) { return model_type_set_ex(S, mt, 1); }
static int model_type_set_ex(MLIS_Ctx* S, int mt, int by_option)
{	/* mlis_model_type_set :738-786: per-model defaults of size and clip_skip */
	int w = 0, skip = 0;
	switch (mt) {
	case MLIS_MODEL_TYPE_NONE: break;
	case MLIS_MODEL_TYPE_SD1: w = 512; skip = 1; break;
	case MLIS_MODEL_TYPE_SD2: w = 768; skip = 2; break;
	case MLIS_MODEL_TYPE_SDXL: w = 1024; skip = 2; break;
	case MLIS_MODEL_TYPE_AMD_TINY: case MLIS_MODEL_TYPE_AMD_TINYV: w = 64; skip = 1; break;
	case MLIS_MODEL_TYPE_AMD_TINYXL: w = 64; skip = 2; break;
	default: return api_error(S, MLIS_E_OPT_VALUE, "invalid model type %d", mt);
	}
	if (mt) { if (S->width <= 0) S->width = w; if (S->height <= 0) S->height = S->width; if (S->clip_skip <= 0) S->clip_skip = skip; }
	S->model_type = mt;
	if (by_option) { if (mt) S->flags |= CF_MODEL_TYPE_SET; else S->flags &= ~CF_MODEL_TYPE_SET; }
	snprintf(S->mname, sizeof(S->mname), "%s", mt ? model_name(mt) : "");
	return 1;
}

static int image_to_tensor(MLIS_Tensor* T, const MLIS_Image* I)
{	/* mlis_tensor_from_image :141-158: u8 HWC -> f32 CHW / 255 */
	const int n0 = (int)I->w, n1 = (int)I->h, n2 = (int)I->c;
	if (!(n0 * n1 * n2 > 0 && I->d)) return MLIS_E_IMAGE;
	mlis_tensor_resize(T, n0, n1, n2, 1);
	const float f = 1 / 255.0;
	for (int y=0;y<n1;++y) for (int x=0;x<n0;++x) for (int c=0;c<n2;++c)
		T->d[(size_t)n0*n1*c + (size_t)n0*y + x] = I->d[(size_t)n0*n2*y + (size_t)n2*x + c] * f;
	return 1;
}

static void prompt_set(MLIS_Ctx* S, int neg, const char* text, int* err)
{
	MLISPrompt *P = neg ? &S->nprompt : &S->prompt;
	str_set(neg ? &S->nprompt_raw : &S->prompt_raw, text);
	S->have_ptok[neg] = 0;
	if (S->flags & CF_NO_PROMPT_PARSE) mlis_prompt_set_raw(P, text);
	else if (mlis_prompt_set_parse(P, text) < 0) { *err = api_error_lib(S, MLIS_E_PROMPT_PARSE); return; }
	else for (int i=0; i<P->n_lora; ++i) {            /* <lora:NAME:MULT> in the prompt (:49-52) */
		int r = lora_add(S, P->lora_names + P->loras[i].name_off, (size_t)P->loras[i].len, P->loras[i].w, LF_PROMPT);
		if (r < 0) { *err = r; return; }
	}
}

static int file_exists(const char* p);

/* mlis_cfg_lora_add + mlis_lora_path_find (:632-680): a path, or NAME -> <lora_dir>/NAME.safetensors */
static int lora_add(MLIS_Ctx* S, const char* name, size_t len, float mult, int flags)
{
	if (S->n_lora >= MAX_LORAS) return api_error(S, MLIS_E_UNKNOWN, "too many loras");
	char path[1200];
	snprintf(path, sizeof(path), "%.*s", (int)len, name);
	if (!file_exists(path)) {
		const char *d = S->path_lora_dir ? S->path_lora_dir : "";
		const size_t dl = strlen(d);
		snprintf(path, sizeof(path), "%s%s%.*s.safetensors", d, (dl && d[dl-1] != '/' && d[dl-1] != '\\') ? "/" : "", (int)len, name);
		if (!file_exists(path)) return api_error(S, MLIS_E_FILE_NOT_FOUND, "lora model file not found '%s'", path);
	}
	S->loras[S->n_lora].path = strdup(path); S->loras[S->n_lora].mult = mult; S->loras[S->n_lora].flags = flags;
	S->n_lora++;
	S->rflags &= ~READY_LORAS;
	return 1;
}

static void loras_remove(MLIS_Ctx* S, int only_prompt)
{
	int k = 0;
	for (int i=0;i<S->n_lora;++i) {
		if (only_prompt && !(S->loras[i].flags & LF_PROMPT)) { S->loras[k++] = S->loras[i]; continue; }
		free(S->loras[i].path);
		S->rflags &= ~READY_LORAS;
	}
	S->n_lora = k;
}

typedef struct { int is_str; va_list* ap; const char* cur; const char* arg_b; size_t arg_n; } ArgSrc;

static void next_str_arg(ArgSrc* A)
{	/* value_str_next :845-864: comma separated, optional double quotes */
	const char *cur = A->cur;
	if (*cur == ',') cur++;
	if (*cur == '"') { cur++; A->arg_b = cur; while (*cur && *cur != '"') cur++; A->arg_n = cur - A->arg_b; if (*cur == '"') cur++; }
	else { A->arg_b = cur; while (*cur && *cur != ',') cur++; A->arg_n = cur - A->arg_b; }
	A->cur = cur;
}
static int arg_int(ArgSrc* A, int mn, int mx, int def, int* out)
{
	int v;
	if (A->is_str) {
		next_str_arg(A);
		char *tail = (char*)A->arg_b + A->arg_n;
		v = A->arg_n ? (int)strtol(A->arg_b, &tail, 10) : def;
		if (tail != A->arg_b + A->arg_n) return 0;
	} else v = va_arg(*A->ap, int);
	if (!(mn <= v && v <= mx)) return 0;
	*out = v; return 1;
}
static int arg_float(ArgSrc* A, float mn, float mx, float def, float* out)
{
	float v;
	if (A->is_str) {
		next_str_arg(A);
		char *tail = (char*)A->arg_b + A->arg_n;
		v = A->arg_n ? strtof(A->arg_b, &tail) : def;
		if (tail != A->arg_b + A->arg_n) return 0;
	} else v = (float)va_arg(*A->ap, double);
	if (!(mn <= v && v <= mx)) return 0;
	*out = v; return 1;
}
static int arg_bool(ArgSrc* A, int* out)
{
	if (!A->is_str) { *out = !!va_arg(*A->ap, int); return 1; }
	next_str_arg(A);
	static const char* const t[] = {"true","yes","y","1"}; static const char* const f[] = {"false","no","n","0"};
	for (int i=0;i<4;++i) { if (A->arg_n == strlen(t[i]) && !memcmp(A->arg_b, t[i], A->arg_n)) { *out = 1; return 1; }
	                        if (A->arg_n == strlen(f[i]) && !memcmp(A->arg_b, f[i], A->arg_n)) { *out = 0; return 1; } }
	return 0;
}
/* whole remaining value (ARG_STR_NO_PARSE) or one comma-separated piece (ARG_STR); returns a malloc'd copy */
static char* arg_str(ArgSrc* A, int whole, size_t mn)
{
	if (!A->is_str) { const char *s = va_arg(*A->ap, const char*); if (!s || strlen(s) < mn) return NULL; return strdup(s); }
	if (whole) { A->arg_b = A->cur; A->arg_n = strlen(A->cur); A->cur += A->arg_n; } else next_str_arg(A);
	if (A->arg_n < mn) return NULL;
	char *r = (char*)malloc(A->arg_n + 1); memcpy(r, A->arg_b, A->arg_n); r[A->arg_n] = 0;
	return r;
}

static int option_apply(MLIS_Ctx* S, int id, ArgSrc* A)
{	/* src/mlimgsynth_options_set.c.h, option by option */
	int i = 0, j = 0, err = 1; float f = 0; char *s = NULL;
#define BAD_VALUE do { free(s); return A->is_str ? api_error(S, MLIS_E_OPT_VALUE, "invalid argument '%.*s' for option '%s'", (int)A->arg_n, A->arg_b ? A->arg_b : "", mlis_option_str(id)) \
	: api_error(S, MLIS_E_OPT_VALUE, "invalid argument for option '%s'", mlis_option_str(id)); } while (0)
#define NO_STR do { if (A->is_str) return api_error(S, MLIS_E_OPT_VALUE, "option '%s' cannot be set with a string value", mlis_option_str(id)); } while (0)
	switch (id) {
	case MLIS_OPT_BACKEND: {
		if (!(s = arg_str(A, 0, 0))) BAD_VALUE;
		char *p2 = A->is_str ? arg_str(A, 0, 0) : NULL;
		if (!A->is_str) { const char *p = va_arg(*A->ap, const char*); (void)p; }
		free(p2);
		str_set(&S->backend, s); S->rflags &= ~READY_BACKEND;
	} break;
	case MLIS_OPT_MODEL: if (!(s = arg_str(A, 1, 1))) BAD_VALUE; str_set(&S->path_model, s); model_drop(S); break;
	case MLIS_OPT_TAE: if (!(s = arg_str(A, 1, 0))) BAD_VALUE; str_set(&S->path_tae, s);
		if (*s) S->flags |= CF_USE_TAE; else S->flags &= ~CF_USE_TAE; engine_drop(S); break;
	case MLIS_OPT_MODEL_TYPE:
		if (A->is_str) { next_str_arg(A); i = model_from(A->arg_b, A->arg_n); if (i < 0) BAD_VALUE; } else i = va_arg(*A->ap, int);
		if (model_type_set(S, i) < 0) return MLIS_E_OPT_VALUE;
		model_drop(S);
		break;
	case MLIS_OPT_AUX_DIR: if (!(s = arg_str(A, 1, 0))) BAD_VALUE; str_set(&S->path_aux, s); if (S->tok) { clip_tokr_free(S->tok); S->tok = NULL; } break;
	case MLIS_OPT_LORA_DIR: if (!(s = arg_str(A, 1, 0))) BAD_VALUE; str_set(&S->path_lora_dir, s); break;
	case MLIS_OPT_LORA: {
		if (!(s = arg_str(A, 0, 1))) BAD_VALUE;
		f = 1;
		if (A->is_str) { if (!arg_float(A, 0, 1, 1, &f)) BAD_VALUE; } else f = (float)va_arg(*A->ap, double);
		err = lora_add(S, s, strlen(s), f, 0);
	} break;
	case MLIS_OPT_LORA_CLEAR: loras_remove(S, 0); break;
	case MLIS_OPT_PROMPT: if (!(s = arg_str(A, 1, 0))) BAD_VALUE; prompt_set(S, 0, s, &err); break;
	case MLIS_OPT_NPROMPT: if (!(s = arg_str(A, 1, 0))) BAD_VALUE; prompt_set(S, 1, s, &err); break;
	case MLIS_OPT_NO_PROMPT_PARSE: if (!arg_bool(A, &i)) BAD_VALUE; if (i) S->flags |= CF_NO_PROMPT_PARSE; else S->flags &= ~CF_NO_PROMPT_PARSE; break;
	case MLIS_OPT_IMAGE_DIM: if (!arg_int(A, 0, 65535, 0, &i) || !arg_int(A, 0, 65535, 0, &j)) BAD_VALUE; S->width = i; S->height = j; break;
	case MLIS_OPT_BATCH_SIZE: if (!arg_int(A, 0, 1024, 0, &i)) BAD_VALUE; S->n_batch = i; break;
	case MLIS_OPT_CLIP_SKIP: if (!arg_int(A, 0, 255, 0, &i)) BAD_VALUE; S->clip_skip = i; break;
	case MLIS_OPT_CFG_SCALE: if (!arg_float(A, 0, 255, NAN, &f)) BAD_VALUE; S->cfg_scale = f; break;
	case MLIS_OPT_METHOD:
		if (A->is_str) {
			next_str_arg(A);
			size_t n = A->arg_n; int anc = 0;
			if (n > 2 && !memcmp(A->arg_b + n - 2, "_a", 2)) { n -= 2; anc = 1; }        /* "euler_a": ancestral shortcut (:88-99) */
			i = from_list(k_method, COUNTOF(k_method), 1, A->arg_b, n);
			if (i < 0) { if (anc) return api_error(S, MLIS_E_OPT_VALUE, "invalid method name '%.*s'", (int)A->arg_n, A->arg_b); BAD_VALUE; }
			if (anc) S->s_ancestral = 1;
		} else i = va_arg(*A->ap, int);
		S->method = i; break;
	case MLIS_OPT_SCHEDULER:
		if (A->is_str) { next_str_arg(A); i = from_list(k_sched, COUNTOF(k_sched), 1, A->arg_b, A->arg_n); if (i < 0) BAD_VALUE; } else i = va_arg(*A->ap, int);
		S->sched = i; break;
	case MLIS_OPT_STEPS: if (!arg_int(A, 0, 1000, 0, &i)) BAD_VALUE; S->n_step = i; break;
	case MLIS_OPT_F_T_INI: if (!arg_float(A, 0, 1, NAN, &f)) BAD_VALUE; S->f_t_ini = f; break;
	case MLIS_OPT_F_T_END: if (!arg_float(A, 0, 1, NAN, &f)) BAD_VALUE; S->f_t_end = f; break;
	case MLIS_OPT_S_NOISE: if (!arg_float(A, 0, 255, NAN, &f)) BAD_VALUE; S->s_noise = f; break;
	case MLIS_OPT_S_ANCESTRAL: if (!arg_float(A, 0, 255, NAN, &f)) BAD_VALUE; S->s_ancestral = f; break;
	case MLIS_OPT_IMAGE: {
		NO_STR;
		const MLIS_Image *img = va_arg(*A->ap, const MLIS_Image*);
		if (!img || (img->c != 3 && img->c != 4)) return api_error(S, MLIS_E_IMAGE, "invalid number of channels in image: %d", img ? (int)img->c : 0);
		if (image_to_tensor(&S->image, img) < 0) return api_error(S, MLIS_E_IMAGE, "invalid image");
		S->tuflags |= MLIS_TUF_IMAGE;
		if (S->image.n[2] == 4) {   /* alpha channel = in-painting mask */
			const size_t wh = (size_t)S->image.n[0] * S->image.n[1];
			mlis_tensor_resize(&S->mask, S->image.n[0], S->image.n[1], 1, 1);
			memcpy(S->mask.d, S->image.d + wh*3, wh*4);
			S->image.n[2] = 3;
			S->tuflags |= MLIS_TUF_MASK;
		}
	} break;
	case MLIS_OPT_IMAGE_MASK: {
		NO_STR;
		const MLIS_Image *img = va_arg(*A->ap, const MLIS_Image*);
		if (!img || img->c != 1) return api_error(S, MLIS_E_IMAGE, "invalid number of channels in image mask: %d", img ? (int)img->c : 0);
		if (image_to_tensor(&S->mask, img) < 0) return api_error(S, MLIS_E_IMAGE, "invalid image mask");
		S->tuflags |= MLIS_TUF_MASK;
	} break;
	case MLIS_OPT_NO_DECODE: if (!arg_bool(A, &i)) BAD_VALUE; if (i) S->flags |= CF_NO_DECODE; else S->flags &= ~CF_NO_DECODE; break;
	case MLIS_OPT_TENSOR_USE_FLAGS: if (!arg_int(A, 0, 0x7fffffff, 0, &i)) BAD_VALUE; S->tuflags = i; break;
	case MLIS_OPT_SEED:
		if (A->is_str) {
			if (!A->cur[0]) break;                                   /* empty string: keep the random seed */
			next_str_arg(A);
			char *tail = NULL; uint64_t v = (uint64_t)strtoll(A->arg_b, &tail, 10);
			if (tail != A->arg_b + A->arg_n) BAD_VALUE;
			S->seed = v;
		} else S->seed = va_arg(*A->ap, uint64_t);
		break;
	case MLIS_OPT_VAE_TILE: if (!arg_int(A, 0, 65535, 0, &i)) BAD_VALUE; S->vae_tile = i; break;
	case MLIS_OPT_UNET_SPLIT: if (!arg_bool(A, &i)) BAD_VALUE; if (i) S->flags |= CF_UNET_SPLIT; else S->flags &= ~CF_UNET_SPLIT; break;   /* weight streaming through three device slabs (engine_get) */
	case MLIS_OPT_WEIGHT_TYPE:
		if (A->is_str) {
			next_str_arg(A);
			static const struct { const char* n; int t; } wt[] = { {"f32", MLT_F32}, {"f16", MLT_F16}, {"bf16", MLT_BF16} };
			int found = 0;
			for (int k=0;k<3;++k) if (id_eq(A->arg_b, A->arg_n, wt[k].n)) { S->wtype = wt[k].t; S->flags |= CF_WEIGHT_TYPE_SET; found = 1; }
			if (found) break;
			char *tail = (char*)A->arg_b + A->arg_n; i = A->arg_n ? (int)strtol(A->arg_b, &tail, 10) : 0;
			if (tail != A->arg_b + A->arg_n) BAD_VALUE;
		} else i = va_arg(*A->ap, int);
		if (i == -1) { S->wtype = MLT_F16; S->flags &= ~CF_WEIGHT_TYPE_SET; }
		else if (i == MLT_F32 || i == MLT_F16 || i == MLT_BF16) { S->wtype = i; S->flags |= CF_WEIGHT_TYPE_SET; }
		else return api_error(S, MLIS_E_OPT_VALUE, "weight type %d is not supported (f32, f16, bf16; quantised ggml types are not)", i);
		break;
	case MLIS_OPT_THREADS: if (!arg_int(A, 0, 65535, 0, &i)) BAD_VALUE; S->n_thread = i; break;
	case MLIS_OPT_DUMP_FLAGS: if (!arg_int(A, 0, 0x7fffffff, 0, &i)) BAD_VALUE; S->dump_flags = i; break;
	case MLIS_OPT_CALLBACK: NO_STR; S->callback = va_arg(*A->ap, MLIS_Callback); S->callback_ud = va_arg(*A->ap, void*); break;
	case MLIS_OPT_ERROR_HANDLER: NO_STR; S->errh = va_arg(*A->ap, MLIS_ErrorHandler); S->errh_ud = va_arg(*A->ap, void*); break;
	case MLIS_OPT_LOG_LEVEL:
		if (A->is_str) { next_str_arg(A); } else (void)va_arg(*A->ap, int);   /* this library does not log: accepted, ignored */
		break;
	default: return api_error(S, MLIS_E_UNK_OPT, "unknown option %u", (unsigned)id);
	}
	free(s);
	return err;
#undef BAD_VALUE
#undef NO_STR
}

MLB_API int mlis_option_set(MLIS_Ctx* S, MLIS_Option id, ...)
{
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	va_list ap; va_start(ap, id);
	ArgSrc A = { 0, &ap, NULL, NULL, 0 };
	int r = option_apply(S, (int)id, &A);
	va_end(ap);
	return r;
}

MLB_API int mlis_option_set_str(MLIS_Ctx* S, const char* name, const char* value)
{
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	int id = name ? (int)mlis_option_fromz(name) : -1;
	if (id <= 0) return api_error(S, MLIS_E_UNK_OPT, "unknown option '%s'", name ? name : "");
	ArgSrc A = { 1, NULL, value ? value : "", NULL, 0 };
	return option_apply(S, id, &A);
}

MLB_API int mlis_option_get(MLIS_Ctx* S, MLIS_Option id, ...)
{	/* src/mlimgsynth_options_get.c.h: the four options the reference implements */
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	va_list ap; va_start(ap, id);
	int r = 1;
	switch ((int)id) {
	case MLIS_OPT_MODEL: { const char **p = va_arg(ap, const char**); if (p) *p = S->path_model ? S->path_model : ""; } break;
	case MLIS_OPT_MODEL_TYPE: { int *p = va_arg(ap, int*); if (p) *p = S->model_type; } break;
	case MLIS_OPT_PROMPT: { const char **p = va_arg(ap, const char**); if (p) *p = S->prompt_raw ? S->prompt_raw : ""; } break;
	case MLIS_OPT_NPROMPT: { const char **p = va_arg(ap, const char**); if (p) *p = S->nprompt_raw ? S->nprompt_raw : ""; } break;
	default: r = api_error(S, MLIS_E_UNK_OPT, "unknown option %u", (unsigned)id);
	}
	va_end(ap);
	return r;
}

MLB_API int mlis_amd_prompt_tokens_set(MLIS_Ctx* S, const int32_t* tokens, const float* weights, int n, int negative)
{
	const int k = negative ? 1 : 0;
	if (n < 0 || (n && !tokens)) return api_error(S, MLIS_E_OPT_VALUE, "invalid token list");
	free(S->ptok[k]); free(S->ptokw[k]);
	S->ptok[k] = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1)); S->ptokw[k] = (float*)malloc(sizeof(float) * (n ? n : 1));
	for (int i=0;i<n;++i) { S->ptok[k][i] = tokens[i]; S->ptokw[k][i] = weights ? weights[i] : 1.0f; }
	S->n_ptok[k] = n; S->have_ptok[k] = 1;
	return 1;
}

/* ------------------------------------------------------------------ setup */
MLB_API const MLIS_BackendInfo* mlis_backend_info_get(MLIS_Ctx* S, unsigned idx, int flags)
{	/* one backend: "HIP" with its devices (mlis_backend_info_get :572-612) */
	(void)flags;
	if (!S || idx != 0) return NULL;
	int n = mlsd_device_count(); if (n > 16) n = 16; if (n < 0) n = 0;
	for (int d=0; d<n; ++d) {
		int ncu = 0; size_t tot = 0, fr = 0;
		mlsd_device_info(d, S->dev_names[d][1], 96, S->dev_names[d][0], 96, &ncu, &tot, &fr);
		char nm[96]; snprintf(nm, sizeof(nm), "HIP%d", d); snprintf(S->dev_names[d][0], 96, "%s", nm);
		S->devs[d].name = S->dev_names[d][0]; S->devs[d].desc = S->dev_names[d][1]; S->devs[d].mem_free = fr; S->devs[d].mem_total = tot;
	}
	S->backend_info.name = "HIP"; S->backend_info.n_dev = (unsigned)n; S->backend_info.devs = S->devs;
	return &S->backend_info;
}

static int file_exists(const char* p) { struct stat st; return p && *p && !stat(p, &st); }

MLB_API int mlis_setup(MLIS_Ctx* S)
{
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	if (!(S->rflags & READY_BACKEND)) {
		/* mlis_backend_init :1119-1161: "" / "HIP" / "HIPn" select a device; anything else is an unknown backend */
		int dev = 0;
		if (!str_empty(S->backend)) {
			if (!strncasecmp(S->backend, "hip", 3) || !strncasecmp(S->backend, "rocm", 4)) dev = atoi(S->backend + (tolower((unsigned char)S->backend[0]) == 'h' ? 3 : 4));
			else return api_error(S, MLIS_E_UNKNOWN, "backend '%s' not available (this library drives MI355X through HIP only)", S->backend);
		}
		if (!mlsd_runtime_is_dry()) {
			if (mlsd_device_count() <= 0) return api_error(S, MLIS_E_UNKNOWN, "no HIP device available (there is no CPU fallback)");
			if (mlsd_device_set(dev)) return api_error_lib(S, MLIS_E_UNKNOWN);
		}
		S->rflags |= READY_BACKEND;
	}
	if (!(S->rflags & READY_MODEL)) {
		if (str_empty(S->path_model)) return api_error(S, MLIS_E_FILE_NOT_FOUND, "no model set (option MODEL)");
		if (S->ts) { mlts_close(S->ts); S->ts = NULL; }               /* left over from a set-up that failed after opening the file */
		if (S->ts_tae) { mlts_close(S->ts_tae); S->ts_tae = NULL; }
		if (!strncmp(S->path_model, "synth:", 6)) {
			char nm[32]; snprintf(nm, sizeof(nm), "%s", S->path_model + 6);
			char *c = strchr(nm, ':'); S->synth_seed = 1234;
			if (c) { *c = 0; S->synth_seed = strtoull(c + 1, NULL, 10); }
			const int mt = model_from(nm, strlen(nm));
			if (mt <= 0) return api_error(S, MLIS_E_OPT_VALUE, "unknown synthetic model '%s'", nm);
			if (model_type_set_ex(S, mt, 0) < 0) return MLIS_E_OPT_VALUE;
			S->synth = 1;
		} else {
			/* mlis_model_load :1163-1204 + mlis_model_identify :1206-1249 */
			S->synth = 0;
			S->ts = mlts_open_safetensors(S->path_model, 1);
			if (!S->ts) return api_error_lib(S, file_exists(S->path_model) ? MLIS_E_UNKNOWN : MLIS_E_FILE_NOT_FOUND);
			int wt = -1;
			const char *m = mlts_model_identify(S->ts, &wt);
			if (m) { if (model_type_set_ex(S, model_from(m, strlen(m)), 0) < 0) { mlts_close(S->ts); S->ts = NULL; return MLIS_E_OPT_VALUE; } }
			else if (!(S->flags & CF_MODEL_TYPE_SET)) { mlts_close(S->ts); S->ts = NULL; return api_error(S, MLIS_E_UNKNOWN, "could not detect the model type"); }
			if (wt >= 0 && !(S->flags & CF_WEIGHT_TYPE_SET)) S->wtype = (wt == MLT_F32 || wt == MLT_BF16) ? wt : MLT_F16;   /* quantised GGUF weights: dequantised, kept as F16 */
		}
		if ((S->flags & CF_USE_TAE) && !S->synth && !str_empty(S->path_tae)) {
			S->ts_tae = mlts_open_safetensors(S->path_tae, 0);
			if (!S->ts_tae) { if (S->ts) { mlts_close(S->ts); S->ts = NULL; } return api_error_lib(S, file_exists(S->path_tae) ? MLIS_E_UNKNOWN : MLIS_E_FILE_NOT_FOUND); }
		}
		S->rflags |= READY_MODEL;
		S->rflags &= ~READY_LORAS;
		S->n_lora_applied = 0;
	}
	if (!(S->rflags & READY_LORAS)) {
		/* :1276-1296: patched tensors are dropped (the store is re-read) and every active LoRA is merged again; the resident
		 * engine and text towers hold the old weights, so they are rebuilt */
		if (S->n_lora || S->n_lora_applied) {
			if (S->synth) return api_error(S, MLIS_E_UNKNOWN, "LoRA needs a checkpoint file (the synthetic models have no tensor store)");
			engine_drop(S); textcond_drop(S);
			if (S->n_lora_applied) {
				mlts_close(S->ts);
				S->ts = mlts_open_safetensors(S->path_model, 1);
				if (!S->ts) { S->rflags &= ~READY_MODEL; return api_error_lib(S, MLIS_E_UNKNOWN); }
			}
			S->n_lora_applied = 0;
			for (int i=0;i<S->n_lora;++i) {
				MLTStore *L = mlts_open_lora(S->loras[i].path);
				if (!L) return api_error_lib(S, MLIS_E_UNKNOWN);
				int r = mlts_lora_apply(S->ts, L, S->loras[i].mult, S->wtype);
				mlts_close(L);
				if (r < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
				S->n_lora_applied++;
			}
		}
		S->rflags |= READY_LORAS;
	}
	return 1;
}

/* TAE files carry bare names ("decoder.layers.N..."): the reference prefixes them with "tae." at load
 * (tensor_callback_prefix_add, src/mlimgsynth.c:1057-1066).  Here: lookup with the prefix stripped. */
static int tae_load(MLIS_Ctx* S, MLCtx* C)
{
	const int np = mlctx_param_count(C);
	for (int i=0;i<np;++i) {
		const char *key; int type; int64_t ne[4];
		mlctx_param_info(C, i, &key, &type, ne);
		const MLTSEntry *e = mlts_find(S->ts_tae, !strncmp(key, "tae.", 4) ? key + 4 : key);
		if (!e) return api_error(S, MLIS_E_UNKNOWN, "tensor '%s' not found", key);
		const int64_t cnt = e->shape[0]*e->shape[1]*e->shape[2]*e->shape[3];
		if (cnt != ne[0]*ne[1]*ne[2]*ne[3]) return api_error(S, MLIS_E_UNKNOWN, "tensor '%s': wrong element count", key);
		if (mlctx_param_set(C, key, e->dtype, e->data, cnt) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
	}
	return 1;
}

static int ctx_weights(MLIS_Ctx* S, MLCtx* C, int is_tae)
{
	if (S->synth) return mlctx_params_synth(C, S->synth_seed) < 0 ? api_error_lib(S, MLIS_E_UNKNOWN) : 1;
	if (is_tae) { if (!S->ts_tae) return api_error(S, MLIS_E_FILE_NOT_FOUND, "no TAE model set (option TAE)"); return tae_load(S, C); }
	return mlctx_tstore_load(C, S->ts) < 0 ? api_error_lib(S, MLIS_E_UNKNOWN) : 1;
}

static int sampler_defaults(MLIS_Ctx* S, int* n_step, int* method, int* sched)
{
	*n_step = S->n_step >= 1 ? S->n_step : 20;                       /* sampling.c:41-42 */
	*method = S->method > 0 ? S->method : MLIS_METHOD_EULER;         /* :33 */
	*sched = S->sched > 0 ? S->sched : MLIS_SCHED_UNIFORM;           /* :69 */
	return 1;
}

/* the engine for (model, size, batch, guidance on/off, codec); rebuilt only when one of these changes */
static int engine_get(MLIS_Ctx* S, int lw, int lh)
{
	const int f = 8, B = S->n_batch > 0 ? S->n_batch : 1, tae = !!(S->flags & CF_USE_TAE);
	char key[96];
	snprintf(key, sizeof(key), "%s/%dx%d/b%d/g%d/t%d/w%d/s%d", S->mname, lw, lh, B, S->cfg_scale > 1, tae, S->wtype, !!(S->flags & CF_UNET_SPLIT));
	int n_step, method, sched;
	sampler_defaults(S, &n_step, &method, &sched);
	if (!S->eng || strcmp(key, S->eng_key)) {
		engine_drop(S);
		MLIS_AmdConfig c; memset(&c, 0, sizeof(c));
		c.model = S->mname; c.width = lw * f; c.height = lh * f; c.n_batch = B; c.n_step = n_step; c.cfg_scale = S->cfg_scale;
		c.s_ancestral = S->s_ancestral; c.sched = sched; c.use_tae = tae; c.weight_seed = S->synth_seed; c.method = method;
		c.s_noise = S->s_noise; c.f_t_ini = S->f_t_ini; c.f_t_end = S->f_t_end; c.defer_weights = 1;
		c.unet_split = (S->flags & CF_UNET_SPLIT) ? 1 : 0;      /* MLIS_OPT_UNET_SPLIT (src/mlimgsynth.c:1629 unet_split): the UNet's weights are streamed, not resident */
		S->eng = mlis_amd_create(&c, NULL);
		if (!S->eng) return api_error_lib(S, MLIS_E_UNKNOWN);
		mlctx_set_wtype(mlis_amd_unet_ctx(S->eng), S->wtype);
		if (ctx_weights(S, mlis_amd_unet_ctx(S->eng), 0) < 0 || ctx_weights(S, mlis_amd_decoder_ctx(S->eng), tae) < 0) { engine_drop(S); return -1; }
		snprintf(S->eng_key, sizeof(S->eng_key), "%s", key);
		if (S->dump_flags & 4) {   /* MLIS_DUMP_GRAPH (src/mlimgsynth.c:432,1298 -> MLB_F_DUMP -> "dump-graph-<name>.txt", src/mlblock.c:111-116) */
			mlctx_block_graph_dump_path(mlis_amd_unet_ctx(S->eng), "dump-graph-unet.txt");
			mlctx_block_graph_dump_path(mlis_amd_decoder_ctx(S->eng), tae ? "dump-graph-tae.txt" : "dump-graph-vae.txt");
		}
	}
	if (mlis_amd_set_sampler(S->eng, n_step, method, sched, S->cfg_scale, S->s_ancestral, S->s_noise, S->f_t_ini, S->f_t_end) < 0)
		return api_error_lib(S, MLIS_E_OPT_VALUE);
	/* VAE_TILE (:1318,1345): tile-sized decoder plan, weights loaded like the full-size one */
	mlis_amd_set_vae_tile(S->eng, S->vae_tile);
	MLCtx *tc = mlis_amd_decoder_tile_prepare(S->eng);
	if (tc && !mlctx_params_loaded(tc) && ctx_weights(S, tc, 0) < 0) return -1;
	return 1;
}

static int textcond_get(MLIS_Ctx* S, int w, int h)
{
	char key[64];
	snprintf(key, sizeof(key), "%s/skip%d", S->mname, S->clip_skip);
	if (!S->tc || strcmp(key, S->tc_key)) {
		textcond_drop(S);
		S->tc = mlis_amd_textcond_create_ex(S->mname, w, h, S->synth_seed, NULL, S->clip_skip, 1);
		if (!S->tc) return api_error_lib(S, MLIS_E_UNKNOWN);
		for (int i=0;i<mlis_amd_textcond_n_towers(S->tc);++i)
			if (ctx_weights(S, mlis_amd_textcond_ctx(S->tc, i), 0) < 0) { textcond_drop(S); return -1; }
		snprintf(S->tc_key, sizeof(S->tc_key), "%s", key);
	}
	mlis_amd_textcond_set_size(S->tc, w, h);
	return 1;
}

/* ------------------------------------------------------------------ text */
static int tokenizer_get(MLIS_Ctx* S)
{
	if (S->tok) return 1;
	static const char* const names[] = { "bpe_simple_vocab_16e6.txt", "merges.txt", "clip_merges.txt" };
	static const char* const dirs[] = { NULL, ".", "/usr/share/mlimgsynth", "/usr/local/share/mlimgsynth" };   /* mlis_file_find :711-735 */
	char path[1024];
	for (int d=0; d<4; ++d) for (int n=0; n<3; ++n) {
		const char *dir = d == 0 ? S->path_aux : dirs[d];
		if (str_empty(dir)) continue;
		snprintf(path, sizeof(path), "%s/%s", dir, names[n]);
		if (!file_exists(path)) continue;
		S->tok = clip_tokr_new();
		if (clip_tokr_load_merges_txt(S->tok, path, 0) < 0) { clip_tokr_free(S->tok); S->tok = NULL; return api_error_lib(S, MLIS_E_UNKNOWN); }
		return 1;
	}
	return api_error(S, MLIS_E_FILE_NOT_FOUND, "CLIP vocabulary not found: put bpe_simple_vocab_16e6.txt (or merges.txt) in the AUX_DIR directory");
}

/* mlis_prompt_text_tokenize :1367-1403: chunks are tokenized one by one, every token carries its chunk's weight */
static int prompt_tokenize(MLIS_Ctx* S, const MLISPrompt* P, int32_t** ptok, float** pw)
{
	int n = 0, cap = 128;
	int32_t *tok = (int32_t*)malloc(sizeof(int32_t) * cap); float *w = (float*)malloc(sizeof(float) * cap);
	int nonempty = 0;
	for (int i=0;i<P->n_chunk;++i) if (P->chunks[i].len > 0) nonempty = 1;
	if (nonempty) { int r = tokenizer_get(S); if (r < 0) { free(tok); free(w); return r; } }
	for (int i=0; i<P->n_chunk && nonempty; ++i) {
		int32_t buf[512];
		int k = clip_tokenize(S->tok, P->text + P->chunks[i].begin, P->chunks[i].len, buf, 512);
		if (k < 0) { free(tok); free(w); return api_error_lib(S, MLIS_E_UNKNOWN); }
		if (n + k > cap) { cap = (n + k) * 2; tok = (int32_t*)realloc(tok, sizeof(int32_t)*cap); w = (float*)realloc(w, sizeof(float)*cap); }
		for (int j=0;j<k;++j) { tok[n+j] = buf[j]; w[n+j] = P->chunks[i].w; }
		n += k;
	}
	*ptok = tok; *pw = w;
	return n;
}

MLB_API int mlis_text_tokenize(MLIS_Ctx* S, const char* text, int32_t** ptokens, MLIS_SubModel model)
{
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	if (model != MLIS_SUBMODEL_CLIP && model != MLIS_SUBMODEL_CLIP2) return api_error(S, MLIS_E_UNKNOWN, "invalid model for text tokenization: %d", model);
	mlis_prompt_set_raw(&S->prompt, text);                            /* prompt_text_set_raw (:1411) */
	free(S->tokens); free(S->tokens_w); S->tokens = NULL; S->tokens_w = NULL;
	int n = prompt_tokenize(S, &S->prompt, &S->tokens, &S->tokens_w);
	if (n < 0) return n;
	S->n_tokens = n;
	if (ptokens) *ptokens = S->tokens;
	return n;
}

MLB_API int mlis_clip_text_encode(MLIS_Ctx* S, const char* text, MLIS_Tensor* embed, MLIS_Tensor* feat, MLIS_SubModel model, int flags)
{	/* :1470-1483 -> mlis_clip_tokens_encode: one tower, clip_skip option, final norm unless MLIS_CTEF_NO_NORM */
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	int r = mlis_setup(S); if (r < 0) return r;
	int32_t *tk = NULL;
	int n = mlis_text_tokenize(S, text, &tk, model); if (n < 0) return n;
	const int xl = !strcmp(S->mname, "sdxl") || !strcmp(S->mname, "tinyxl");
	if (model == MLIS_SUBMODEL_CLIP2 && !xl) return api_error(S, MLIS_E_UNKNOWN, "invalid model for text tokenize: %d", model);
	const char *tower = model == MLIS_SUBMODEL_CLIP2 ? (S->mname[0] == 's' ? "vit_bigg" : "tiny")
		: (!strcmp(S->mname, "sd2") ? "vit_h" : (S->mname[0] == 's' ? "vit_l" : "tiny"));
	ClipParams P; if (clip_params_get(tower, &P) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
	MLCtx *C = mlctx_new(NULL);
	ClipEncoder E;
	r = clip_encoder_init(&E, C, &P, model == MLIS_SUBMODEL_CLIP2 ? "clip2" : "clip", 1, S->clip_skip, !(flags & MLIS_CTEF_NO_NORM), feat != NULL);
	if (r > 0) r = ctx_weights(S, C, 0); else api_error_lib(S, MLIS_E_UNKNOWN);
	if (r > 0) {
		if (embed) mlis_tensor_resize(embed, P.d_embed, P.n_token, 1, 1);
		if (feat) mlis_tensor_resize(feat, P.d_embed, 1, 1, 1);
		r = clip_encoder_run(&E, (unsigned)n, tk, embed ? embed->d : NULL, feat ? feat->d : NULL);
		if (r < 0) api_error_lib(S, MLIS_E_UNKNOWN);
	}
	clip_encoder_free(&E); mlctx_destroy(C);
	return r < 0 ? r : 1;
}

/* tokens + emphasis weights of the prompt (neg = 0) or the negative prompt: explicit ids or the parsed prompt text */
static int prompt_tokens(MLIS_Ctx* S, int neg, int32_t** ptok, float** ptw)
{
	if (S->have_ptok[neg]) {
		const int n = S->n_ptok[neg];
		*ptok = (int32_t*)malloc(sizeof(int32_t)*(n ? n : 1)); *ptw = (float*)malloc(sizeof(float)*(n ? n : 1));
		memcpy(*ptok, S->ptok[neg], sizeof(int32_t)*n); memcpy(*ptw, S->ptokw[neg], sizeof(float)*n);
		return n;
	}
	MLISPrompt *P = neg ? &S->nprompt : &S->prompt;
	if (!P->n_chunk) mlis_prompt_set_raw(P, "");
	return prompt_tokenize(S, P, ptok, ptw);
}

/* mlis_text_cond_encode :1501-1563 for the prompt and (with_neg) the negative prompt: one run per tower for both */
static int text_cond_encode(MLIS_Ctx* S, int with_neg, int w, int h)
{
	int32_t *tok[2] = {NULL, NULL}; float *tw[2] = {NULL, NULL}; int n[2] = {0, 0};
	int r = -1;
	if ((n[0] = prompt_tokens(S, 0, &tok[0], &tw[0])) < 0) { r = n[0]; goto end; }
	if (with_neg && (n[1] = prompt_tokens(S, 1, &tok[1], &tw[1])) < 0) { r = n[1]; goto end; }
	if (textcond_get(S, w, h) < 0) goto end;
	int n_ctx = 0, n_label = 0;
	mlis_amd_textcond_dims(S->tc, &n_ctx, &n_label);
	mlis_tensor_resize(&S->cond, n_ctx, 77, 1, 1);
	if (n_label) mlis_tensor_resize(&S->label, n_label, 1, 1, 1);
	if (with_neg) {
		mlis_tensor_resize(&S->ncond, n_ctx, 77, 1, 1);
		if (n_label) mlis_tensor_resize(&S->nlabel, n_label, 1, 1, 1);
		r = mlis_amd_textcond_encode_pair_w(S->tc, tok[0], tw[0], n[0], tok[1], tw[1], n[1], S->cond.d, n_label ? S->label.d : NULL,
			S->ncond.d, n_label ? S->nlabel.d : NULL);
	} else r = mlis_amd_textcond_encode_w(S->tc, tok[0], tw[0], n[0], S->cond.d, n_label ? S->label.d : NULL);
	if (r < 0) r = api_error_lib(S, MLIS_E_UNKNOWN);
end:
	free(tok[0]); free(tw[0]); free(tok[1]); free(tw[1]);
	return r < 0 ? r : 1;
}

/* ------------------------------------------------------------------ codec */
static int progress(MLIS_Ctx* S, MLIS_Stage stage, int step, int step_end)
{	/* mlis_callback :614-629 */
	const double t = now_s();
	S->prg.stage = stage; S->prg.step = step; S->prg.step_end = step_end; S->prg.step_time = t - S->t_last; S->prg.time = t;
	S->t_last = t;
	if (S->callback) { int r = S->callback(S->callback_ud, S, &S->prg); if (r < 0) return r; }
	return 1;
}

MLB_API int mlis_mask_encode(MLIS_Ctx* S, const MLIS_Tensor* mask, MLIS_Tensor* lmask, int flags)
{	/* ltensor_downsize(lmask, mask, f, f, 1, 1): box average (src/localtensor.c:161-194) */
	(void)flags;
	const int f = 8, w = mask->n[0], h = mask->n[1], lw = w / f, lh = h / f;
	float *out = (float*)malloc(sizeof(float) * ((size_t)lw*lh + 1));
	const float fn = 1.0f / (f*f);
	for (int i1=0;i1<lh;++i1) for (int i0=0;i0<lw;++i0) {
		float v = 0;
		for (int j1=0;j1<f;++j1) for (int j0=0;j0<f;++j0) v += mask->d[i0*f + j0 + (size_t)(i1*f + j1)*w];
		out[i0 + (size_t)i1*lw] = v * fn;
	}
	mlis_tensor_resize(lmask, lw, lh, 1, 1);
	memcpy(lmask->d, out, sizeof(float) * (size_t)lw * lh);
	free(out);
	(void)S;
	return 1;
}

static int engine_for_image(MLIS_Ctx* S, int w, int h)
{
	if (w % 8 || h % 8 || w < 8 || h < 8) return api_error(S, MLIS_E_IMAGE, "invalid input image shape: %dx%d", w, h);
	return engine_get(S, w / 8, h / 8);
}

MLB_API int mlis_image_encode(MLIS_Ctx* S, const MLIS_Tensor* image, MLIS_Tensor* latent, int flags)
{	/* :1301-1330; batch: the same image for every batch element */
	(void)flags;
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	int r = mlis_setup(S); if (r < 0) return r;
	if (image->n[2] != 3 || image->n[3] != 1) return api_error(S, MLIS_E_IMAGE, "invalid input image shape: %dx%dx%dx%d", image->n[0], image->n[1], image->n[2], image->n[3]);
	const int w = image->n[0], h = image->n[1];
	if (engine_for_image(S, w, h) < 0) return -1;
	MLCtx *ec = mlis_amd_encoder_tile_prepare(S->eng);          /* VAE_TILE: the tile-sized encoder, when tiling applies */
	if (!ec) ec = mlis_amd_encoder_ctx(S->eng);
	if (!ec) ec = mlis_amd_encoder_prepare(S->eng);
	if (!ec) return api_error_lib(S, MLIS_E_UNKNOWN);
	if (!mlctx_params_loaded(ec) && ctx_weights(S, ec, !!(S->flags & CF_USE_TAE)) < 0) return -1;
	const int B = S->n_batch > 0 ? S->n_batch : 1;
	const size_t per = (size_t)3 * w * h;
	float *imgs = (float*)malloc(per * B * 4);
	for (int b=0;b<B;++b) memcpy(imgs + per*b, image->d, per*4);
	uint64_t seeds[MAX_IMAGES]; for (int b=0;b<B && b<MAX_IMAGES;++b) seeds[b] = S->seed + b;
	mlis_amd_seed_ex(S->eng, seeds, S->rng_offset);
	r = mlis_amd_encode(S->eng, imgs, 1);
	free(imgs);
	if (r < 0) return api_error_lib(S, r == MLIS_E_NAN ? MLIS_E_NAN : MLIS_E_UNKNOWN);
	S->rng_offset = mlis_amd_rng_offset(S->eng);
	mlis_tensor_resize(latent, w/8, h/8, 4, 1);
	if (mlsd_memcpy(latent->d, mlis_amd_latent_device(S->eng), (size_t)4*(w/8)*(h/8)*4, 1, NULL) || mlsd_device_sync()) return api_error_lib(S, MLIS_E_UNKNOWN);
	return progress(S, MLIS_STAGE_IMAGE_ENCODE, 1, 1);
}

static int image_fetch(MLIS_Ctx* S, MLIS_Tensor* image, int w, int h)
{
	const int B = S->n_batch > 0 ? S->n_batch : 1;
	mlis_tensor_resize(image, w, h, 3, B);
	if (mlis_amd_sync(S->eng) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);        /* the decode was enqueued on the engine's own stream */
	if (mlsd_memcpy(image->d, mlis_amd_image_device(S->eng), (size_t)B*3*w*h*4, 1, NULL) || mlsd_device_sync()) return api_error_lib(S, MLIS_E_UNKNOWN);
	const size_t n = mlis_tensor_count(image);
	for (size_t i=0;i<n;++i) if (!isfinite(image->d[i])) return api_error(S, MLIS_E_NAN, "NaN found in decoded image");   /* :1348-1349 */
	image->flags |= LT_F_READY;
	return 1;
}

MLB_API int mlis_image_decode(MLIS_Ctx* S, const MLIS_Tensor* latent, MLIS_Tensor* image, int flags)
{	/* :1332-1357; the latent of batch element 0 is replicated when the tensor holds one image */
	(void)flags;
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	int r = mlis_setup(S); if (r < 0) return r;
	if (latent->n[2] != 4) return api_error(S, MLIS_E_UNKNOWN, "latent must have 4 channels");
	const int lw = latent->n[0], lh = latent->n[1], B = S->n_batch > 0 ? S->n_batch : 1;
	if (engine_get(S, lw, lh) < 0) return -1;
	const size_t per = (size_t)4*lw*lh;
	float *l = (float*)malloc(per * B * 4);
	for (int b=0;b<B;++b) memcpy(l + per*b, latent->d + (latent->n[3] == B ? per*b : 0), per*4);
	r = mlis_amd_set_init_latent(S->eng, l);
	free(l);
	if (r < 0 || mlis_amd_decode(S->eng) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
	mlis_amd_set_init_latent(S->eng, NULL);
	if (image_fetch(S, image, lw*8, lh*8) < 0) return -1;
	return progress(S, MLIS_STAGE_IMAGE_DECODE, 1, 1);
}

/* ------------------------------------------------------------------ generation */
static int denoise_cb(void* user, int step, int n_step, int nfe)
{
	MLIS_Ctx *S = (MLIS_Ctx*)user;
	S->prg.nfe = nfe;
	return progress(S, MLIS_STAGE_DENOISE, step, n_step);
}

static void infotext_update(MLIS_Ctx* S, int w, int h)
{	/* mlis_infotext_update :1589-1632 (imitates stable-diffusion-webui create_infotext) */
	char buf[4096]; int n = 0;
#define ADD(...) n += snprintf(buf + n, sizeof(buf) - n > 0 ? sizeof(buf) - n : 0, __VA_ARGS__)
	int n_step, method, sched; sampler_defaults(S, &n_step, &method, &sched);
	ADD("%s\n", S->prompt_raw ? S->prompt_raw : "");
	if (!str_empty(S->nprompt_raw)) ADD("Negative prompt: %s\n", S->nprompt_raw);
	ADD("Seed: %" PRIu64, S->seed);
	ADD(", Sampler: %s", mlis_method_str((MLIS_Method)method));
	if (S->s_ancestral == 1) ADD(" ancestral");
	ADD(", Schedule type: %s", mlis_sched_str((MLIS_Scheduler)sched));
	if (S->s_ancestral > 0) ADD(", Ancestral: %g", S->s_ancestral);
	if (S->s_noise > 0) ADD(", SNoise: %g", S->s_noise);
	if (S->cfg_scale > 1) ADD(", CFG scale: %g", S->cfg_scale);
	if (S->f_t_ini < 1) ADD(", Mode: %s, f_t_ini: %g", tensor_good(&S->lmask) ? "inpaint" : "img2img", S->f_t_ini);
	ADD(", Steps: %u", (unsigned)S->last_n_step);
	ADD(", NFE: %u", (unsigned)S->last_nfe);
	ADD(", Size: %ux%u", (unsigned)w, (unsigned)h);
	ADD(", Clip skip: %d", S->clip_skip);
	{
		const char *b = S->path_model ? S->path_model : "", *sl = strrchr(b, '/'); if (sl) b = sl + 1;
		const char *e = strrchr(b, '.'); if (!e) e = b + strlen(b);
		ADD(", Model: %.*s", (int)(e - b), b);
	}
	if (S->flags & CF_USE_TAE) ADD(", VAE: tae");
	ADD(", Version: MLImgSynth v%s", MLIS_VERSION_STR);
#undef ADD
	free(S->infotext); S->infotext = strdup(buf);
}

MLB_API int mlis_generate(MLIS_Ctx* S)
{
	if (!S || S->signature != CTX_SIGNATURE) return -1;
	int r = mlis_setup(S); if (r < 0) return r;
	const int B = S->n_batch > 0 ? S->n_batch : 1;
	if (B > MAX_IMAGES) return api_error(S, MLIS_E_OPT_VALUE, "batch size > %d not supported", MAX_IMAGES);
	S->t_last = now_s(); memset(&S->prg, 0, sizeof(S->prg));
	const double t_start = S->t_last; (void)t_start;
	int w = S->width / 8, h = S->height / 8;

	/* img2img source (:1652-1657) */
	if (S->tuflags & MLIS_TUF_IMAGE) {
		if ((r = mlis_image_encode(S, &S->image, &S->latent, 0)) < 0) return r;
		S->tuflags |= MLIS_TUF_LATENT;
	}
	if (S->tuflags & MLIS_TUF_LATENT) { w = S->latent.n[0]; h = S->latent.n[1]; }
	if (w < 1 || h < 1) return api_error(S, MLIS_E_OPT_VALUE, "image size not set");
	if (engine_get(S, w, h) < 0) return -1;
	if (S->tuflags & MLIS_TUF_LATENT) {
		if (S->latent.n[2] != 4) return api_error(S, MLIS_E_UNKNOWN, "latent must have 4 channels");
		const size_t per = (size_t)4*w*h;
		float *l = (float*)malloc(per * B * 4);
		for (int b=0;b<B;++b) memcpy(l + per*b, S->latent.d + (S->latent.n[3] == B ? per*b : 0), per*4);
		r = mlis_amd_set_init_latent(S->eng, l);
		free(l);
		if (r < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
	} else mlis_amd_set_init_latent(S->eng, NULL);
	const int w_img = w * 8, h_img = h * 8;

	/* mask -> latent mask (:1673-1686) */
	if (S->tuflags & MLIS_TUF_MASK) { mlis_mask_encode(S, &S->mask, &S->lmask, 0); S->tuflags |= MLIS_TUF_LMASK; }
	if ((S->tuflags & MLIS_TUF_LMASK) && tensor_good(&S->lmask)) {
		if (S->lmask.n[0] != w || S->lmask.n[1] != h) return api_error(S, MLIS_E_IMAGE, "latent mask %dx%d does not match the latent %dx%d", S->lmask.n[0], S->lmask.n[1], w, h);
		if (mlis_amd_set_lmask(S->eng, S->lmask.d) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
	} else { mlis_amd_set_lmask(S->eng, NULL); if (!(S->tuflags & MLIS_TUF_LMASK)) mlis_tensor_free(&S->lmask); }

	/* conditioning (:1688-1707) */
	if (!(S->tuflags & MLIS_TUF_CONDITIONING)) {
		if ((r = text_cond_encode(S, S->cfg_scale > 1, w_img, h_img)) < 0) return r;
		if (S->cfg_scale > 1) {
			UnetParams U; unet_params_get(S->mname, &U);
			const int neg_empty = S->have_ptok[1] ? S->n_ptok[1] == 0 : str_empty(S->nprompt_raw);
			if (U.uncond_empty_zero && neg_empty) memset(S->ncond.d, 0, mlis_tensor_count(&S->ncond) * 4);   /* :1702-1703 */
		}
		if ((r = progress(S, MLIS_STAGE_COND_ENCODE, 1, 1)) < 0) return r;
	}
	if (!tensor_good(&S->cond) || (S->cfg_scale > 1 && !tensor_good(&S->ncond))) return api_error(S, MLIS_E_UNKNOWN, "conditioning tensors are not set");
	if (mlis_amd_set_cond(S->eng, S->cond.d, tensor_good(&S->label) ? S->label.d : NULL, S->cfg_scale > 1 ? S->ncond.d : NULL,
			(S->cfg_scale > 1 && tensor_good(&S->nlabel)) ? S->nlabel.d : NULL) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
	S->image.flags &= ~LT_F_READY;

	/* sampling (:1720-1748): image i of the batch has its own Philox stream seed + i, continuing at the context's offset */
	uint64_t seeds[MAX_IMAGES]; for (int b=0;b<B;++b) seeds[b] = S->seed + b;
	mlis_amd_seed_ex(S->eng, seeds, S->rng_offset);
	mlis_amd_set_callback(S->eng, S->callback ? denoise_cb : NULL, S);
	r = mlis_amd_denoise(S->eng, NULL);
	S->rng_offset = mlis_amd_rng_offset(S->eng);
	S->last_n_step = mlis_amd_last_n_step(S->eng); S->last_nfe = mlis_amd_last_nfe(S->eng); S->prg.nfe = S->last_nfe;
	if (r < 0) {
		if (r < -1 && r != MLIS_E_NAN) return r;                                  /* the callback's abort code */
		return api_error_lib(S, r == MLIS_E_NAN || strstr(mlsd_last_error(), "NaN") ? MLIS_E_NAN : MLIS_E_UNKNOWN);
	}
	mlis_tensor_resize(&S->latent, w, h, 4, B);
	if (mlsd_memcpy(S->latent.d, mlis_amd_latent_device(S->eng), (size_t)B*4*w*h*4, 1, NULL) || mlsd_device_sync()) return api_error_lib(S, MLIS_E_UNKNOWN);

	/* decode (:1752-1756) */
	if (!(S->flags & CF_NO_DECODE)) {
		if (mlis_amd_decode(S->eng) < 0) return api_error_lib(S, MLIS_E_UNKNOWN);
		if ((r = image_fetch(S, &S->image, w_img, h_img)) < 0) return r;
		if ((r = progress(S, MLIS_STAGE_IMAGE_DECODE, 1, 1)) < 0) return r;
	}
	infotext_update(S, w_img, h_img);
	/* mlis_prompt_clear :692-709 */
	str_set(&S->prompt_raw, ""); str_set(&S->nprompt_raw, "");
	mlis_prompt_free(&S->prompt); mlis_prompt_free(&S->nprompt);
	S->have_ptok[0] = S->have_ptok[1] = 0;
	S->f_t_ini = 1; S->f_t_end = 0; S->tuflags = 0;
	loras_remove(S, 1);                                  /* mlis_cfg_loras_prompt_remove */
	return 1;
}

MLB_API MLIS_Image* mlis_image_get(MLIS_Ctx* S, int idx)
{	/* :1775-1793 + mlis_tensor_to_image :118-139: u8 = clamp(v*255, 0, 255) truncated */
	if (!S || S->signature != CTX_SIGNATURE) return NULL;
	if (!(S->image.flags & LT_F_READY)) { api_error(S, MLIS_E_UNKNOWN, "image not ready"); return NULL; }
	if (idx < 0 || idx >= S->image.n[3] || idx >= MAX_IMAGES) { api_error(S, MLIS_E_UNKNOWN, "only image idx < %d available", S->image.n[3]); return NULL; }
	const int n0 = S->image.n[0], n1 = S->image.n[1], n2 = S->image.n[2];
	MLIS_Image *I = &S->imgex[idx];
	I->w = n0; I->h = n1; I->c = n2; I->sz = (size_t)n0*n1*n2;
	I->d = (uint8_t*)realloc((I->flags & LT_F_OWNMEM) ? I->d : NULL, I->sz ? I->sz : 1);
	I->flags |= LT_F_OWNMEM;
	const float *td = S->image.d + (size_t)n0*n1*n2*idx;
	for (int y=0;y<n1;++y) for (int x=0;x<n0;++x) for (int c=0;c<n2;++c) {
		float v = td[(size_t)n0*n1*c + (size_t)n0*y + x] * 255;
		v = v < 0 ? 0 : (v > 255 ? 255 : v);
		I->d[(size_t)n0*n2*y + (size_t)n2*x + c] = (uint8_t)v;
	}
	return I;
}

MLB_API const char* mlis_infotext_get(MLIS_Ctx* S, int idx)
{
	if (!S || S->signature != CTX_SIGNATURE || idx < 0) return NULL;
	return S->infotext;
}

MLB_API MLIS_Tensor* mlis_tensor_get(MLIS_Ctx* S, MLIS_TensorId id)
{	/* :1795-1820 */
	if (!S || S->signature != CTX_SIGNATURE) return NULL;
	switch ((int)id) {
	case MLIS_TENSOR_IMAGE: return &S->image;   case MLIS_TENSOR_MASK: return &S->mask;
	case MLIS_TENSOR_LATENT: return &S->latent; case MLIS_TENSOR_LMASK: return &S->lmask;
	case MLIS_TENSOR_COND: return &S->cond;     case MLIS_TENSOR_LABEL: return &S->label;
	case MLIS_TENSOR_NCOND: return &S->ncond;   case MLIS_TENSOR_NLABEL: return &S->nlabel;
	}
	if ((int)id >= MLIS_TENSOR_TMP && (int)id < MLIS_TENSOR_TMP + N_TMP_TENSORS) return &S->tmp[(int)id - MLIS_TENSOR_TMP];
	api_error(S, MLIS_E_UNKNOWN, "invalid tensor id %d", (int)id);
	return NULL;
}
