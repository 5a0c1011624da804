/* Checkpoint index + parameter loading — re-creation, for the hot path, of the reference's tensor store
 * (src/ccompute/tensorstore.c:184-323, tensorstore_safet.c:152-205) and of the load callbacks of src/mlimgsynth.c:
 *   tensor_callback_main :1033-1055  every file tensor is renamed to the engine's dotted names (name_conv.c = tnconv_sd);
 *   open_clip_attn_conv  :989-1030   open_clip's fused attn.in_proj_{weight,bias} becomes three views q/k/v_proj;
 *   mlis_model_identify  :1206-1249  model type + linear weight type from one cross-attention k_proj tensor;
 *   mlctx_tstore_load    src/mlblock.c:266-292  every plan parameter is looked up by name, the element COUNT is
 *                        checked (not the shape, :243: that is how SDXL's Linear proj_in/out load into 1x1 convs),
 *                        converted to the parameter's type and uploaded (here: repacked to the device layout).
 * The file is mmap'd read-only; nothing is copied until a parameter is uploaded.  Two containers, detected by content like
 * tstore_read's format probe: safetensors (JSON header) and GGUF v2/v3 (tensorstore_gguf.c:171-235: key/value metadata is
 * skipped, tensor data starts at the next multiple of 32 after the tensor table).  GGUF tensors may be block-quantised
 * (Q8_0, Q4_0, Q4_1, Q5_0, Q5_1): the reference converts them through ggml's type traits (tensorstore.c:187-225), which
 * are not in the reference tree; the block layouts below restate ggml's published formats (ggml-common.h block_q*),
 * dequantised to fp32 on upload (the device weight type is F16 either way).  PARITY of the quantised types is UNPINNED.
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <ctype.h>
#include <math.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct MLTStore {
	MLTSEntry* e; int n, cap;
	void** owned; int n_owned, cap_owned;     /* heap buffers of patched tensors (LoRA) */
	void* map; size_t map_size;
	int n_unused, n_split;
};

/* ------------------------------------------------------------------ tiny JSON cursor (safetensors header only) */
typedef struct { const char *p, *end; } JCur;

static void j_ws(JCur* c) { while (c->p < c->end && isspace((unsigned char)*c->p)) c->p++; }
static int j_eat(JCur* c, char ch) { j_ws(c); if (c->p < c->end && *c->p == ch) { c->p++; return 1; } return 0; }

static int j_string(JCur* c, char* out, size_t max)
{	/* JSON string with the escapes a tensor name can contain; returns length or -1 */
	j_ws(c);
	if (c->p >= c->end || *c->p != '"') return -1;
	c->p++;
	size_t n = 0;
	while (c->p < c->end && *c->p != '"') {
		char ch = *c->p++;
		if (ch == '\\' && c->p < c->end) {
			ch = *c->p++;
			if (ch == 'n') ch = '\n'; else if (ch == 't') ch = '\t';
			else if (ch == 'u') { if (c->end - c->p < 4) return -1; c->p += 4; ch = '?'; }
		}
		if (n + 1 < max) out[n++] = ch;
	}
	if (c->p >= c->end) return -1;
	c->p++;
	out[n] = 0;
	return (int)n;
}

static int j_uint(JCur* c, uint64_t* v)
{
	j_ws(c);
	if (c->p >= c->end || !isdigit((unsigned char)*c->p)) return -1;
	uint64_t x = 0;
	while (c->p < c->end && isdigit((unsigned char)*c->p)) x = x*10 + (uint64_t)(*c->p++ - '0');
	*v = x;
	return 1;
}

static int j_skip(JCur* c)
{	/* skip one value */
	j_ws(c);
	if (c->p >= c->end) return -1;
	if (*c->p == '"') { char tmp[8]; return j_string(c, tmp, sizeof(tmp)) < 0 ? -1 : 1; }
	if (*c->p == '{' || *c->p == '[') {
		const char open = *c->p, close = open == '{' ? '}' : ']';
		c->p++;
		if (j_eat(c, close)) return 1;
		do {
			if (open == '{') { char tmp[8]; if (j_string(c, tmp, sizeof(tmp)) < 0 || !j_eat(c, ':')) return -1; }
			if (j_skip(c) < 0) return -1;
		} while (j_eat(c, ','));
		return j_eat(c, close) ? 1 : -1;
	}
	while (c->p < c->end && *c->p != ',' && *c->p != '}' && *c->p != ']') c->p++;
	return 1;
}

/* ------------------------------------------------------------------ store */
static int dtype_from_str(const char* s)
{
	if (!strcmp(s, "F32")) return MLT_F32;
	if (!strcmp(s, "F16")) return MLT_F16;
	if (!strcmp(s, "BF16")) return MLT_BF16;
	if (!strcmp(s, "F64")) return MLT_F64;
	if (!strcmp(s, "I64")) return MLT_I64;
	if (!strcmp(s, "I32")) return MLT_I32;
	return -1;
}

static size_t dtype_size(int t)
{
	switch (t) { case MLT_F16: case MLT_BF16: return 2; case MLT_F32: case MLT_I32: return 4; case MLT_F64: case MLT_I64: return 8; }
	return 0;
}

/* block-quantised ggml types: elements per block / bytes per block (0 = not a block type) */
static int qblock_elems(int t) { return (t == MLT_Q8_0 || t == MLT_Q4_0 || t == MLT_Q4_1 || t == MLT_Q5_0 || t == MLT_Q5_1) ? 32 : 0; }
static size_t qblock_bytes(int t)
{
	switch (t) { case MLT_Q8_0: return 34; case MLT_Q4_0: return 18; case MLT_Q4_1: return 20; case MLT_Q5_0: return 22; case MLT_Q5_1: return 24; }
	return 0;
}
/* bytes of n elements of type t, or 0 if n is not a whole number of blocks / the type is unknown */
static size_t dtype_bytes(int t, int64_t n)
{
	const int qb = qblock_elems(t);
	if (qb) return (n % qb) ? 0 : (size_t)(n / qb) * qblock_bytes(t);
	return (size_t)n * dtype_size(t);
}

/* ggml block formats (restated from the published layout; little-endian, fp16 scales):
 *   Q8_0: d, q[32] int8                      x[j] = d*q[j]
 *   Q4_0: d, qs[16]                          x[j] = d*((qs[j]&15) - 8),  x[j+16] = d*((qs[j]>>4) - 8)
 *   Q4_1: d, m, qs[16]                       x[j] = d*(qs[j]&15) + m,    x[j+16] = d*(qs[j]>>4) + m
 *   Q5_0: d, qh[4], qs[16]                   5th bit of x[j] = bit j of qh, of x[j+16] = bit j+16;   x = d*(q - 16)
 *   Q5_1: d, m, qh[4], qs[16]                x = d*q + m */
static void dequant_blocks(int t, const unsigned char* b, int64_t n, float* o)
{
	const size_t bs = qblock_bytes(t);
	for (int64_t ib=0; ib<n/32; ++ib, b+=bs, o+=32) {
		uint16_t dh; memcpy(&dh, b, 2);
		const float d = mlb_f16_bits_to_f32(dh);
		if (t == MLT_Q8_0) { const signed char *q = (const signed char*)b + 2; for (int j=0;j<32;++j) o[j] = d * (float)q[j]; continue; }
		float m = 0.f; const unsigned char *q = b + 2;
		if (t == MLT_Q4_1 || t == MLT_Q5_1) { uint16_t mh; memcpy(&mh, q, 2); m = mlb_f16_bits_to_f32(mh); q += 2; }
		uint32_t qh = 0;
		if (t == MLT_Q5_0 || t == MLT_Q5_1) { memcpy(&qh, q, 4); q += 4; }
		for (int j=0;j<16;++j) {
			int x0 = q[j] & 15, x1 = q[j] >> 4;
			if (t == MLT_Q5_0 || t == MLT_Q5_1) { x0 |= (int)((qh >> j) & 1) << 4; x1 |= (int)((qh >> (j + 16)) & 1) << 4; }
			if (t == MLT_Q4_0) { o[j] = d * (float)(x0 - 8); o[j+16] = d * (float)(x1 - 8); }
			else if (t == MLT_Q5_0) { o[j] = d * (float)(x0 - 16); o[j+16] = d * (float)(x1 - 16); }
			else { o[j] = d * (float)x0 + m; o[j+16] = d * (float)x1 + m; }
		}
	}
}

/* element count of a shape read from a file, or -1 if a dimension or the product is out of range (corrupt headers must not
 * overflow the size arithmetic: no tensor of a checkpoint has more than 2^40 elements) */
static int64_t shape_count(const uint64_t* dims, int nd)
{
	uint64_t n = 1;
	for (int i=0;i<nd;++i) {
		if (dims[i] > 0xffffffffull) return -1;
		if (dims[i] && n > ((uint64_t)1 << 40) / dims[i]) return -1;
		n *= dims[i];
	}
	return (int64_t)n;
}

static MLTSEntry* ts_add(MLTStore* S, const char* name, const MLTSEntry* src)
{
	for (int i=0;i<S->n;++i) if (!strcmp(S->e[i].name, name)) {   /* tstore_tensor_add replaces an existing key */
		char *keep = S->e[i].name; S->e[i] = *src; S->e[i].name = keep; return &S->e[i];
	}
	if (S->n == S->cap) { S->cap = S->cap ? S->cap*2 : 1024; S->e = (MLTSEntry*)realloc(S->e, sizeof(MLTSEntry)*S->cap); }
	MLTSEntry *e = &S->e[S->n++];
	*e = *src;
	e->name = strdup(name);
	return e;
}

MLB_API void mlts_close(MLTStore* S)
{
	if (!S) return;
	for (int i=0;i<S->n;++i) free(S->e[i].name);
	for (int i=0;i<S->n_owned;++i) free(S->owned[i]);
	free(S->owned);
	free(S->e);
	if (S->map) munmap(S->map, S->map_size);
	free(S);
}

/* open_clip_attn_conv (src/mlimgsynth.c:989-1030): "<...>attn.in_proj_weight" [d, 3d] -> q/k/v_proj.weight views */
static int qkv_split(MLTStore* S, const MLTSEntry* e, const char* newname)
{
	const char *type;
	size_t l = strlen(newname);
	if (l >= 12 && !strcmp(newname + l - 12, "in_proj_bias")) { type = "bias"; l -= 12; }
	else if (l >= 14 && !strcmp(newname + l - 14, "in_proj_weight")) { type = "weight"; l -= 14; }
	else return 0;
	const int idim = e->shape[1] == 1 ? 0 : 1;
	if (e->shape[idim] % 3) return mlsd_set_error(-1, "invalid open_clip tensor '%s'", newname);
	MLTSEntry part = *e;
	part.shape[idim] /= 3;
	part.size /= 3;
	static const char* const pn[3] = {"q_proj.", "k_proj.", "v_proj."};
	for (int i=0;i<3;++i) {
		char nm[256];
		snprintf(nm, sizeof(nm), "%.*s%s%s", (int)l, newname, pn[i], type);
		ts_add(S, nm, &part);
		part.data = (const char*)part.data + part.size;
	}
	S->n_split++;
	return 1;
}

static MLTStore* open_store(const char* path, int mode);

/* mlts_open: safetensors or GGUF, by content.  convert_names = 1 applies tnconv_sd + the open_clip QKV split
 * (tensor_callback_main); 0 keeps the file's names (internal dotted names: synthetic checkpoints, tests). */
MLB_API MLTStore* mlts_open(const char* path, int convert_names) { return open_store(path, convert_names ? 1 : 0); }
MLB_API MLTStore* mlts_open_safetensors(const char* path, int convert_names) { return open_store(path, convert_names ? 1 : 0); }
/* LoRA file (kohya naming): "lora_" prefix stripped, then tnconv_sd; an unmatched "*.lora_down.weight" is an error, other
 * unmatched tensors are dropped (tensor_callback_lora, src/mlimgsynth.c:1068-1092) */
MLB_API MLTStore* mlts_open_lora(const char* path) { return open_store(path, 2); }

/* one tensor of the file: renamed / split / dropped according to `mode` (0 raw, 1 model, 2 LoRA); <0 on error */
static int add_file_tensor(MLTStore* S, const char* name, const MLTSEntry* e, int mode)
{
	if (!mode) { ts_add(S, name, e); return 1; }
	char conv[512];
	const char *nm = name;
	if (mode == 2) {                                                     /* tensor_callback_lora :1068-1092 */
		if (strncmp(name, "lora_", 5)) { S->n_unused++; return 0; }
		nm = name + 5;
	}
	const int r = tnconv_sd(nm, conv, sizeof(conv));                     /* tensor_callback_main :1033-1055 */
	if (r < 0) return -1;
	if (mode == 2 && r == 0) {
		const size_t l = strlen(nm);
		if (l >= 17 && !strcmp(nm + l - 17, ".lora_down.weight")) return mlsd_set_error(-1, "unmatched lora tensor: %s", name);
		S->n_unused++; return 0;
	}
	if (r == 0) { S->n_unused++; return 0; }
	if (r == TNCONV_R_QKV_PROJ) return qkv_split(S, e, conv) < 0 ? -1 : 1;
	ts_add(S, conv, e);
	return 1;
}

static int parse_safetensors(MLTStore* S, const char* path, int mode)
{
	const unsigned char *b = (const unsigned char*)S->map;
	uint64_t hlen = 0;
	for (int i=7;i>=0;--i) hlen = (hlen << 8) | b[i];
	if (hlen < 2 || hlen > 0xffffff || 8 + hlen > S->map_size || b[8] != '{')   /* tstore_detect_safet :300-308 */
		return mlsd_set_error(-1, "'%s': invalid safetensors header", path);
	const char *data0 = (const char*)b + 8 + hlen;
	const size_t data_size = S->map_size - 8 - hlen;
	JCur c = { (const char*)b + 8, (const char*)b + 8 + hlen };
	char name[512], key[64], val[64];
	if (!j_eat(&c, '{')) goto bad;
	if (!j_eat(&c, '}')) do {
		if (j_string(&c, name, sizeof(name)) < 0 || !j_eat(&c, ':')) goto bad;
		if (!strcmp(name, "__metadata__")) { if (j_skip(&c) < 0) goto bad; continue; }
		MLTSEntry e; memset(&e, 0, sizeof(e));
		uint64_t off0 = 0, off1 = 0, shp[8]; int nd = 0, have = 0;
		if (!j_eat(&c, '{')) goto bad;
		do {
			if (j_string(&c, key, sizeof(key)) < 0 || !j_eat(&c, ':')) goto bad;
			if (!strcmp(key, "dtype")) { if (j_string(&c, val, sizeof(val)) < 0) goto bad; e.dtype = dtype_from_str(val); have |= 1; }
			else if (!strcmp(key, "shape")) {
				if (!j_eat(&c, '[')) goto bad;
				if (!j_eat(&c, ']')) { do { if (nd >= 8 || j_uint(&c, &shp[nd]) < 0) goto bad; nd++; } while (j_eat(&c, ',')); if (!j_eat(&c, ']')) goto bad; }
				have |= 2;
			}
			else if (!strcmp(key, "data_offsets")) {
				if (!j_eat(&c, '[') || j_uint(&c, &off0) < 0 || !j_eat(&c, ',') || j_uint(&c, &off1) < 0 || !j_eat(&c, ']')) goto bad;
				have |= 4;
			}
			else return mlsd_set_error(-1, "safetensors tensor '%s': unknown key '%s'", name, key);
		} while (j_eat(&c, ','));
		if (!j_eat(&c, '}') || have != 7) goto bad;
		if (nd > 4) return mlsd_set_error(-1, "safetensors tensor '%s': %d dimensions", name, nd);
		if (shape_count(shp, nd) < 0) return mlsd_set_error(-1, "safetensors tensor '%s': shape out of range", name);
		if (off1 < off0 || off1 > data_size) return mlsd_set_error(-1, "safetensors tensor '%s': invalid offsets", name);
		e.n_dim = nd;
		for (int i=0;i<4;++i) e.shape[i] = 1;
		for (int i=0;i<nd;++i) e.shape[i] = (int64_t)shp[nd-1-i];           /* reversed: shape[0] fastest (ggml order) */
		e.size = off1 - off0; e.data = data0 + off0;
		const size_t want = (size_t)(e.shape[0]*e.shape[1]*e.shape[2]*e.shape[3]) * dtype_size(e.dtype);
		if (e.dtype < 0 || want != e.size) return mlsd_set_error(-1, "safetensors tensor '%s': invalid size %zu for dtype/shape", name, e.size);
		if (add_file_tensor(S, name, &e, mode) < 0) return -1;
	} while (j_eat(&c, ','));
	if (!j_eat(&c, '}')) goto bad;
	return 1;
bad:
	return mlsd_set_error(-1, "'%s': malformed safetensors header near byte %ld", path, (long)(c.p - (const char*)b));
}

/* ---- GGUF v2 / v3 (tensorstore_gguf.c).  Cursor over the mapped file; every read is bounds-checked. */
typedef struct { const unsigned char *p, *end; } GCur;
static int g_take(GCur* c, void* out, size_t n) { if ((size_t)(c->end - c->p) < n) return -1; memcpy(out, c->p, n); c->p += n; return 0; }
static int g_skip(GCur* c, uint64_t n) { if ((uint64_t)(c->end - c->p) < n) return -1; c->p += n; return 0; }
static int g_string(GCur* c, char* out, size_t max, uint64_t limit)     /* u64 length + bytes; out may be NULL (skipped) */
{
	uint64_t len;
	if (g_take(c, &len, 8) < 0 || len > limit) return -1;
	if (out) { if (len >= max || (uint64_t)(c->end - c->p) < len) return -1; memcpy(out, c->p, len); out[len] = 0; }
	return g_skip(c, len);
}
/* metadata value sizes by gguf type id: u8 i8 u16 i16 u32 i32 f32 bool string array u64 i64 f64 (gguf_meta_type_to_any :22-37) */
static const int k_gguf_scalar_size[13] = { 1, 1, 2, 2, 4, 4, 4, 1, -1, -2, 8, 8, 8 };
static int g_skip_value(GCur* c, uint32_t type, int depth)
{
	if (type >= 13) return -1;
	const int sz = k_gguf_scalar_size[type];
	if (sz > 0) return g_skip(c, (uint64_t)sz);
	if (sz == -1) return g_string(c, NULL, 0, 0xffffff);
	if (depth) return -1;                                                /* arrays of arrays: TS_E_METADATA in the reference */
	uint32_t at; uint64_t len;
	if (g_take(c, &at, 4) < 0 || g_take(c, &len, 8) < 0 || len > 0xffffff || at >= 13) return -1;
	if (k_gguf_scalar_size[at] > 0) return g_skip(c, len * (uint64_t)k_gguf_scalar_size[at]);
	if (k_gguf_scalar_size[at] != -1) return -1;
	for (uint64_t i=0;i<len;++i) if (g_string(c, NULL, 0, 0xffff) < 0) return -1;
	return 0;
}

static int gguf_dtype(uint32_t t)
{
	switch (t) {
	case 0: return MLT_F32; case 1: return MLT_F16; case 30: return MLT_BF16; case 28: return MLT_F64; case 26: return MLT_I32; case 27: return MLT_I64;
	case 8: return MLT_Q8_0; case 2: return MLT_Q4_0; case 3: return MLT_Q4_1; case 6: return MLT_Q5_0; case 7: return MLT_Q5_1;
	}
	return -1;
}

static int parse_gguf(MLTStore* S, const char* path, int mode)
{
	GCur c = { (const unsigned char*)S->map, (const unsigned char*)S->map + S->map_size };
	uint32_t magic, version; uint64_t n_tensor, n_meta;
	if (g_take(&c, &magic, 4) < 0 || g_take(&c, &version, 4) < 0) return mlsd_set_error(-1, "'%s': truncated GGUF header", path);
	if (version != 2 && version != 3) return mlsd_set_error(-1, "'%s': unsupported GGUF version %u", path, version);   /* :190-191 */
	if (g_take(&c, &n_tensor, 8) < 0 || g_take(&c, &n_meta, 8) < 0 || n_tensor > 65535 || n_meta > 65535)
		return mlsd_set_error(-1, "'%s': invalid GGUF counts", path);
	char name[512];
	for (uint64_t i=0;i<n_meta;++i) {
		uint32_t type;
		if (g_string(&c, name, sizeof(name), 256) < 0 || !name[0] || g_take(&c, &type, 4) < 0 || g_skip_value(&c, type, 0) < 0)
			return mlsd_set_error(-1, "'%s': malformed GGUF metadata entry %llu", path, (unsigned long long)i);
	}
	/* the tensor table comes first, the data offset is only known after it: collect, then resolve */
	MLTSEntry *tmp = (MLTSEntry*)calloc((size_t)n_tensor + 1, sizeof(MLTSEntry));
	char (*names)[512] = (char(*)[512])malloc(((size_t)n_tensor + 1) * 512);
	int R = 1;
	for (uint64_t i=0;i<n_tensor && R>0;++i) {
		uint32_t nd, gt; uint64_t dims[4] = {1,1,1,1}, off;
		if (g_string(&c, names[i], 512, 256) < 0 || !names[i][0] || g_take(&c, &nd, 4) < 0 || nd > 4 || g_take(&c, dims, 8*(size_t)nd) < 0 ||
		    g_take(&c, &gt, 4) < 0 || g_take(&c, &off, 8) < 0) { R = mlsd_set_error(-1, "'%s': malformed GGUF tensor entry %llu", path, (unsigned long long)i); break; }
		for (int d=0;d<4;++d) if (dims[d] > 0xffffff) R = mlsd_set_error(-1, "gguf tensor '%s': dimension overflow", names[i]);
		if (R > 0 && shape_count(dims, 4) < 0) R = mlsd_set_error(-1, "gguf tensor '%s': shape out of range", names[i]);
		if (R < 0) break;
		MLTSEntry *e = &tmp[i];
		e->dtype = gguf_dtype(gt);
		if (e->dtype < 0) { R = mlsd_set_error(-1, "gguf tensor '%s': unknown tensor type %u", names[i], gt); break; }
		e->n_dim = (int)nd;
		for (int d=0;d<4;++d) e->shape[d] = (int64_t)dims[d];                /* GGUF dims are already fastest-first */
		e->size = dtype_bytes(e->dtype, e->shape[0]*e->shape[1]*e->shape[2]*e->shape[3]);
		if (!e->size && dims[0]*dims[1]*dims[2]*dims[3] != 0) { R = mlsd_set_error(-1, "gguf tensor '%s': %lld elements are not whole blocks", names[i], (long long)(dims[0]*dims[1]*dims[2]*dims[3])); break; }
		e->data = (const void*)(uintptr_t)off;                             /* relative for now */
	}
	if (R > 0) {
		uint64_t base = (uint64_t)(c.p - (const unsigned char*)S->map);
		base += (32 - base % 32) % 32;                                     /* gguf_align: GGUF_ALIGNMENT 32, general.alignment is not consulted */
		for (uint64_t i=0;i<n_tensor && R>0;++i) {
			MLTSEntry *e = &tmp[i];
			const uint64_t off = base + (uint64_t)(uintptr_t)e->data;
			if (off > S->map_size || e->size > S->map_size - off) { R = mlsd_set_error(-1, "gguf tensor '%s': data outside the file", names[i]); break; }
			e->data = (const char*)S->map + off;
			if (add_file_tensor(S, names[i], e, mode) < 0) R = -1;
		}
	}
	free(names); free(tmp);
	return R;
}

static MLTStore* open_store(const char* path, int mode)
{
	int fd = open(path, O_RDONLY);
	if (fd < 0) { mlsd_set_error(-6 /* MLIS_E_FILE_NOT_FOUND */, "could not open '%s'", path); return NULL; }
	struct stat st;
	if (fstat(fd, &st) < 0 || st.st_size < 10) { close(fd); mlsd_set_error(-1, "'%s': not a safetensors / GGUF file", path); return NULL; }
	void *map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (map == MAP_FAILED) { mlsd_set_error(-1, "mmap of '%s' failed", path); return NULL; }
	MLTStore *S = (MLTStore*)calloc(1, sizeof(*S));
	S->map = map; S->map_size = (size_t)st.st_size;
	const int r = !memcmp(map, "GGUF", 4) ? parse_gguf(S, path, mode) : parse_safetensors(S, path, mode);   /* tstore_detect_gguf :237-242 */
	if (r < 0) { mlts_close(S); return NULL; }
	return S;
}

MLB_API int mlts_count(const MLTStore* S) { return S ? S->n : 0; }
MLB_API const MLTSEntry* mlts_at(const MLTStore* S, int i) { return (S && i >= 0 && i < S->n) ? &S->e[i] : NULL; }
MLB_API int mlts_stats(const MLTStore* S, int* n_unused, int* n_split)
{
	if (n_unused) *n_unused = S->n_unused;
	if (n_split) *n_split = S->n_split;
	return 1;
}

MLB_API const MLTSEntry* mlts_find(const MLTStore* S, const char* name)
{
	for (int i=0; S && i<S->n; ++i) if (!strcmp(S->e[i].name, name)) return &S->e[i];
	return NULL;
}

/* mlis_model_identify, src/mlimgsynth.c:1206-1249: returns the model name ("sd1" | "sd2" | "sdxl") or NULL, and the
 * checkpoint's linear weight type */
MLB_API const char* mlts_model_identify(const MLTStore* S, int* wtype)
{
	const MLTSEntry *te;
	const char *m = NULL;
	if ((te = mlts_find(S, "unet.in.1.1.transf.0.attn2.k_proj.weight"))) {
		if (te->shape[0] == 768) m = "sd1"; else if (te->shape[0] == 1024) m = "sd2";
	} else if ((te = mlts_find(S, "unet.in.4.1.transf.0.attn2.k_proj.weight"))) {
		if (te->shape[0] == 2048) m = "sdxl";
	}
	if (te && wtype) *wtype = te->dtype;
	if (!m) mlsd_set_error(-1, "could not detect the model type");
	return m;
}

/* mlctx_tstore_load, src/mlblock.c:266-292.  `optional_prefix`: parameters under it may be absent (a checkpoint without
 * the VAE encoder, say); everything else must be present.  Returns the number of parameters loaded. */
MLB_API int mlctx_tstore_load(MLCtx* C, const MLTStore* S)
{
	int n = 0;
	const int np = mlctx_param_count(C);
	for (int i=0;i<np;++i) {
		const char *key; int type; int64_t ne[4];
		mlctx_param_info(C, i, &key, &type, ne);
		const MLTSEntry *e = mlts_find(S, key);
		if (!e) return mlsd_set_error(-1, "tensor '%s' not found", key);                 /* mlblock.c:276-277 */
		const int64_t cnt = e->shape[0]*e->shape[1]*e->shape[2]*e->shape[3];
		if (cnt != ne[0]*ne[1]*ne[2]*ne[3])                                              /* tstore_tensor_read :243 */
			return mlsd_set_error(-1, "tensor '%s': %lld elements in the file, %lld expected", key, (long long)cnt,
				(long long)(ne[0]*ne[1]*ne[2]*ne[3]));
		if (qblock_elems(e->dtype)) {                                                    /* data_convert through to_float, tensorstore.c:205-214 */
			float *f = (float*)malloc(sizeof(float) * (size_t)cnt);
			dequant_blocks(e->dtype, (const unsigned char*)e->data, cnt, f);
			const int r = mlctx_param_set(C, key, MLT_F32, f, cnt);
			free(f);
			if (r < 0) return -1;
		}
		else if (mlctx_param_set(C, key, e->dtype, e->data, cnt) < 0) return -1;
		n++;
	}
	return n;
}

/* ------------------------------------------------------------------ LoRA merge (src/lora.c:9-138)
 * For every "<X>.lora_down.weight" of the LoRA store: W[X.weight] += scale * up . down with
 *   down [n_inner][n0], up [n1][n_inner] (row-major as stored), W [n1][n0];  scale = (X.scale | X.alpha / n_inner | 1) * mult.
 * Like the reference the operands are taken at the model's weight type (F16: rounded), the product accumulates in fp32 and
 * the sum is stored back at the weight type; the patched tensor replaces the file's data in the store (heap buffer). */
static float f16_to_f32_(uint16_t h) { return mlb_f16_bits_to_f32(h); }

static float* entry_to_f32(const MLTSEntry* e, int round_f16)
{
	const int64_t n = e->shape[0]*e->shape[1]*e->shape[2]*e->shape[3];
	float *o = (float*)malloc(sizeof(float) * (size_t)(n ? n : 1));
	if (qblock_elems(e->dtype)) {
		dequant_blocks(e->dtype, (const unsigned char*)e->data, n, o);
		if (round_f16) for (int64_t i=0;i<n;++i) o[i] = f16_to_f32_(mlb_f32_to_f16_bits(o[i]));
		return o;
	}
	for (int64_t i=0;i<n;++i) {
		float v;
		switch (e->dtype) {
		case MLT_F16: v = f16_to_f32_(((const uint16_t*)e->data)[i]); break;
		case MLT_BF16: { uint32_t u = (uint32_t)((const uint16_t*)e->data)[i] << 16; memcpy(&v, &u, 4); } break;
		case MLT_F64: { double d; memcpy(&d, (const char*)e->data + i*8, 8); v = (float)d; } break;
		default: memcpy(&v, (const char*)e->data + i*4, 4); break;
		}
		o[i] = round_f16 ? f16_to_f32_(mlb_f32_to_f16_bits(v)) : v;
	}
	return o;
}

/* tstore_tensor_data_get with conversion to F32 (tensorstore.c:256-323): n = element count of the entry */
MLB_API int mlts_entry_to_f32(const MLTSEntry* e, float* out, int64_t n)
{
	if (!e || !out || n != e->shape[0]*e->shape[1]*e->shape[2]*e->shape[3]) return mlsd_set_error(-1, "mlts_entry_to_f32: bad arguments");
	if (e->dtype == MLT_I32 || e->dtype == MLT_I64) return mlsd_set_error(-1, "mlts_entry_to_f32: integer tensor '%s'", e->name);
	float *f = entry_to_f32(e, 0);
	memcpy(out, f, sizeof(float) * (size_t)n);
	free(f);
	return 1;
}

MLB_API int mlts_lora_apply(MLTStore* D, const MLTStore* L, float mult, int wtype)
{
	int n_applied = 0;
	const int r16 = wtype != MLT_F32;
	char key[600];
	for (int i=0; i<L->n; ++i) {
		const MLTSEntry *ld = &L->e[i];
		const size_t ln = strlen(ld->name);
		if (ln < 17 || strcmp(ld->name + ln - 17, ".lora_down.weight")) continue;
		const int bl = (int)(ln - 17);
		snprintf(key, sizeof(key), "%.*s.weight", bl, ld->name);
		MLTSEntry *dst = (MLTSEntry*)mlts_find(D, key);
		if (!dst) return mlsd_set_error(-1, "lora tensor not found in model: %s", key);
		snprintf(key, sizeof(key), "%.*s.lora_up.weight", bl, ld->name);
		const MLTSEntry *lu = mlts_find(L, key);
		if (!lu) return mlsd_set_error(-1, "lora up tensor not found: %s", key);
		snprintf(key, sizeof(key), "%.*s.scale", bl, ld->name);
		const MLTSEntry *ls = mlts_find(L, key);
		snprintf(key, sizeof(key), "%.*s.alpha", bl, ld->name);
		const MLTSEntry *la = mlts_find(L, key);
		/* the file is untrusted input (the reference's arithmetic here is unchecked, src/lora.c:30-50): a 0-dimensional or
		 * zero-sized tensor must not index shape[-1] or divide by zero, and the element counts must factor exactly */
		if (ld->n_dim < 2 || lu->n_dim < 2 || dst->n_dim < 2 || ld->n_dim > 4 || lu->n_dim > 4)
			return mlsd_set_error(-1, "lora up/down invalid shapes for %.*s", bl, ld->name);
		const int64_t n_inner = ld->shape[ld->n_dim - 1];
		const int64_t cd = ld->shape[0]*ld->shape[1]*ld->shape[2]*ld->shape[3], cu = lu->shape[0]*lu->shape[1]*lu->shape[2]*lu->shape[3];
		const int64_t cw = dst->shape[0]*dst->shape[1]*dst->shape[2]*dst->shape[3];
		if (n_inner <= 0 || cd <= 0 || cu <= 0 || cw <= 0 || cd % n_inner || cu % n_inner)
			return mlsd_set_error(-1, "lora up/down invalid shapes for %.*s", bl, ld->name);
		const int64_t n0 = cd / n_inner, n1 = cu / n_inner;
		if (!(ld->n_dim == dst->n_dim && lu->n_dim == dst->n_dim && cw == n0 * n1))
			return mlsd_set_error(-1, "lora up/down invalid shapes for %.*s", bl, ld->name);
		if (ld->dtype == MLT_I32 || ld->dtype == MLT_I64 || lu->dtype == MLT_I32 || lu->dtype == MLT_I64 || dst->dtype == MLT_I32 || dst->dtype == MLT_I64)
			return mlsd_set_error(-1, "lora: integer tensor in %.*s", bl, ld->name);
		float scale = 1;
		if (ls && ls->shape[0]*ls->shape[1]*ls->shape[2]*ls->shape[3] < 1) ls = NULL;
		if (la && la->shape[0]*la->shape[1]*la->shape[2]*la->shape[3] < 1) la = NULL;
		if (ls) { float *t = entry_to_f32(ls, 0); scale = t[0]; free(t); }
		else if (la) { float *t = entry_to_f32(la, 0); scale = t[0] / n_inner; free(t); }
		scale *= mult;
		float *down = entry_to_f32(ld, r16), *up = entry_to_f32(lu, r16), *w = entry_to_f32(dst, r16);
		float *delta = (float*)malloc(sizeof(float) * (size_t)n0);
		for (int64_t o=0;o<n1;++o) {       /* up.down accumulated in fp32, scaled, then added to W (src/lora.c:57-61) */
			for (int64_t c=0;c<n0;++c) delta[c] = 0;
			for (int64_t r=0;r<n_inner;++r) {
				const float u = up[o*n_inner + r];
				const float *dr = down + r*n0;
				for (int64_t c=0;c<n0;++c) delta[c] += u * dr[c];
			}
			float *wr = w + o*n0;
			for (int64_t c=0;c<n0;++c) wr[c] += delta[c] * scale;
		}
		free(delta);
		free(down); free(up);
		{	/* (the reference checks element 0 only, src/lora.c:63; a corrupt adapter can poison any row) */
			int bad = 0;
			for (int64_t c=0; c<cw && !bad; ++c) bad = !isfinite(w[c]);
			if (bad) { free(w); return mlsd_set_error(-1, "NaN in LoRA result"); }
		}
		void *buf;
		if (r16) {
			uint16_t *h = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)cw);
			for (int64_t c=0;c<cw;++c) h[c] = mlb_f32_to_f16_bits(w[c]);
			free(w); buf = h; dst->dtype = MLT_F16; dst->size = (size_t)cw * 2;
		} else { buf = w; dst->dtype = MLT_F32; dst->size = (size_t)cw * 4; }
		dst->data = buf;
		if (D->n_owned == D->cap_owned) { D->cap_owned = D->cap_owned ? D->cap_owned*2 : 64; D->owned = (void**)realloc(D->owned, sizeof(void*)*D->cap_owned); }
		D->owned[D->n_owned++] = buf;
		n_applied++;
	}
	return n_applied;
}
