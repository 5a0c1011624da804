/* CLIP byte-pair-encoding tokenizer (host, integer work).
 *
 * Behaviour follows the reference's clip_tokenize (src/clip.c:59-278; public entry mlis_text_tokenize,
 * include/mlimgsynth.h) and is pinned by the 14 known-answer tests of src/test_text_tokenize_clip.c:41-66
 * (tests/test_tokenizer_cpu.py):
 *   - token ids: 0..255 = bytes in CLIP's bytes_to_unicode order, 256..511 = the same with the end-of-word mark,
 *     512+i = merge i, then start/end (n_vocab = 512 + n_merges + 2 = 49408 for CLIP's 48894 merges);
 *   - words: skip ASCII/Unicode whitespace; a word is a contraction ('s 't 're 've 'm 'll, ASCII case-insensitive;
 *     the reference's list has 've twice and no 'd, kept), or a maximal run of one class: letters, numbers
 *     (a RUN of digits is one word here, unlike OpenAI's single-digit rule: "2025" -> 17 15 17 276), or
 *     anything else that is not whitespace;
 *   - each word is lower-cased per code point, UTF-8 bytes -> byte tokens, last token += 256, then the
 *     lowest-ranked adjacent merge is applied repeatedly (leftmost wins ties).
 * Unlike the reference, the merge table is DATA supplied at run time (the reference compiles a 634 KB table in):
 * either id pairs (clip_tokr_set_merges) or OpenAI's public bpe_simple_vocab_16e6.txt / merges.txt
 * (clip_tokr_load_merges_txt).  Lookup is an open-addressing hash on (left,right) instead of a sorted index. */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include "unicode_tables.h"
#include <stdint.h>

struct ClipTokenizer {
	int32_t *pairs;         /* [n][2] */
	int n;
	uint64_t *hk;           /* hash keys (left<<32 | right), 0 = empty (no merge has left = right = 0 ... guarded) */
	int32_t *hv;            /* merge index */
	uint32_t hmask;
};

/* ---- Unicode helpers (tables generated from Python unicodedata, tools/gen_unicode_tables.py) */
static int uc_category(uint32_t cp)   /* 'L', 'N', 'Z' or 'P' (everything else) */
{
	int lo = 0, hi = UC_N_RANGES - 1;
	while (lo <= hi) {
		int mid = (lo + hi) >> 1;
		if (cp < g_uc_ranges[mid].lo) hi = mid - 1;
		else if (cp > g_uc_ranges[mid].hi) lo = mid + 1;
		else return g_uc_ranges[mid].cat;
	}
	return 'P';
}

static uint32_t uc_lower(uint32_t cp)
{
	if (cp < 128) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
	int lo = 0, hi = UC_N_LOWER - 1;
	while (lo <= hi) {
		int mid = (lo + hi) >> 1;
		if (cp < g_uc_lower[mid].from) hi = mid - 1;
		else if (cp > g_uc_lower[mid].from) lo = mid + 1;
		else return g_uc_lower[mid].to;
	}
	return cp;
}

/* decode one code point; malformed bytes decode as themselves (one byte) so that arbitrary input terminates */
static uint32_t utf8_next(const char** pc, const char* end)
{
	const unsigned char *c = (const unsigned char*)*pc;
	const long left = end - *pc;
	uint32_t cp = c[0]; int n = 1;
	if (c[0] >= 0xF0 && c[0] < 0xF8 && left >= 4) { cp = ((c[0] & 7u) << 18) | ((c[1] & 63u) << 12) | ((c[2] & 63u) << 6) | (c[3] & 63u); n = 4; }
	else if (c[0] >= 0xE0 && c[0] < 0xF0 && left >= 3) { cp = ((c[0] & 15u) << 12) | ((c[1] & 63u) << 6) | (c[2] & 63u); n = 3; }
	else if (c[0] >= 0xC0 && c[0] < 0xE0 && left >= 2) { cp = ((c[0] & 31u) << 6) | (c[1] & 63u); n = 2; }
	*pc += n;
	return cp;
}

static int utf8_put(char* b, uint32_t cp)
{
	if (cp < 0x80) { b[0] = (char)cp; return 1; }
	if (cp < 0x800) { b[0] = (char)(0xC0 | (cp >> 6)); b[1] = (char)(0x80 | (cp & 63)); return 2; }
	if (cp < 0x10000) { b[0] = (char)(0xE0 | (cp >> 12)); b[1] = (char)(0x80 | ((cp >> 6) & 63)); b[2] = (char)(0x80 | (cp & 63)); return 3; }
	b[0] = (char)(0xF0 | (cp >> 18)); b[1] = (char)(0x80 | ((cp >> 12) & 63)); b[2] = (char)(0x80 | ((cp >> 6) & 63)); b[3] = (char)(0x80 | (cp & 63));
	return 4;
}

static int is_ascii_space(uint32_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

/* ---- byte <-> token (CLIP's bytes_to_unicode order: printable bytes first, the rest appended) */
MLB_API int clip_tokr_byte_to_token(int byte)
{
	const int b = byte & 255;
	if (b <= 32) return b + 188;
	if (b <= 126) return b - 33;
	if (b <= 160) return b + 94;
	if (b <= 172) return b - 67;
	if (b == 173) return 255;
	return b - 68;
}

MLB_API int clip_tokr_token_to_byte(int tok)
{
	tok &= 255;
	if (tok <= 93) return tok + 33;
	if (tok <= 105) return tok + 67;
	if (tok <= 187) return tok + 68;
	if (tok <= 220) return tok - 188;
	if (tok <= 254) return tok - 94;
	return 173;
}

/* ---- merge table */
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
#define PAIR_KEY(l, r) ((((uint64_t)(uint32_t)(l)) << 32 | (uint32_t)(r)) + 1)   /* +1: key 0 means empty */

MLB_API ClipTokenizer* clip_tokr_new(void) { return calloc(1, sizeof(ClipTokenizer)); }

static void tokr_clear(ClipTokenizer* T) { free(T->pairs); free(T->hk); free(T->hv); memset(T, 0, sizeof(*T)); }

MLB_API void clip_tokr_free(ClipTokenizer* T) { if (T) { tokr_clear(T); free(T); } }

MLB_API int clip_tokr_set_merges(ClipTokenizer* T, const int32_t* pairs, int n)
{
	if (!T || n < 0 || (n && !pairs)) return mlsd_set_error(-1, "clip_tokr_set_merges: bad arguments");
	tokr_clear(T);
	uint32_t cap = 16;
	while (cap < (uint32_t)n * 2u) cap <<= 1;
	T->pairs = malloc(sizeof(int32_t) * 2 * (n ? n : 1));
	T->hk = calloc(cap, sizeof(uint64_t));
	T->hv = malloc(cap * sizeof(int32_t));
	if (!T->pairs || !T->hk || !T->hv) { tokr_clear(T); return mlsd_set_error(-1, "clip_tokr_set_merges: out of memory"); }
	T->hmask = cap - 1; T->n = n;
	for (int i=0;i<n;++i) {
		const int32_t l = pairs[2*i], r = pairs[2*i+1];
		if (l < 0 || r < 0 || l >= 512 + i || r >= 512 + i) { tokr_clear(T); return mlsd_set_error(-1, "clip merges: pair %d (%d,%d) refers to a token not yet defined", i, l, r); }
		T->pairs[2*i] = l; T->pairs[2*i+1] = r;
		const uint64_t k = PAIR_KEY(l, r);
		uint32_t h = (uint32_t)mix64(k) & T->hmask;
		while (T->hk[h]) {
			if (T->hk[h] == k) { tokr_clear(T); return mlsd_set_error(-1, "clip merges: pair %d (%d,%d) is a duplicate", i, l, r); }
			h = (h + 1) & T->hmask;
		}
		T->hk[h] = k; T->hv[h] = i;
	}
	return 1;
}

MLB_API int clip_tokr_n_merges(const ClipTokenizer* T) { return T ? T->n : 0; }
MLB_API int clip_tokr_n_vocab(const ClipTokenizer* T) { return T ? 512 + T->n + 2 : 0; }

/* token of the merge (left,right), or INT32_MAX */
static int32_t merge_rank(const ClipTokenizer* T, int32_t l, int32_t r)
{
	if (!T->n) return INT32_MAX;
	const uint64_t k = PAIR_KEY(l, r);
	uint32_t h = (uint32_t)mix64(k) & T->hmask;
	while (T->hk[h]) {
		if (T->hk[h] == k) return 512 + T->hv[h];
		h = (h + 1) & T->hmask;
	}
	return INT32_MAX;
}

/* ---- OpenAI vocabulary text format: header line, then "<sym> <sym>" per merge, symbols spelled in the
 * bytes_to_unicode alphabet with "</w>" closing a word.  Symbol -> id through a string hash built as we go. */
typedef struct { char* pool; size_t used, cap; uint32_t *off; int32_t *id; uint32_t mask; } SymMap;

static uint64_t str_hash(const char* s, size_t n) { uint64_t h = 0xcbf29ce484222325ULL; for (size_t i=0;i<n;++i) { h ^= (unsigned char)s[i]; h *= 0x100000001b3ULL; } return mix64(h); }

static int sym_find(const SymMap* M, const char* s, size_t n)
{
	uint32_t h = (uint32_t)str_hash(s, n) & M->mask;
	while (M->off[h]) {
		const char *k = M->pool + M->off[h];
		if (strlen(k) == n && !memcmp(k, s, n)) return M->id[h];
		h = (h + 1) & M->mask;
	}
	return -1;
}

static int sym_add(SymMap* M, const char* s, size_t n, int32_t id)
{
	if (M->used + n + 1 > M->cap) {
		size_t nc = M->cap * 2 + n + 1;
		char *np = realloc(M->pool, nc);
		if (!np) return -1;
		M->pool = np; M->cap = nc;
	}
	uint32_t h = (uint32_t)str_hash(s, n) & M->mask;
	while (M->off[h]) h = (h + 1) & M->mask;
	memcpy(M->pool + M->used, s, n); M->pool[M->used + n] = 0;
	M->off[h] = (uint32_t)M->used; M->id[h] = id;
	M->used += n + 1;
	return 0;
}

/* spelling of byte token t (< 256) in the vocabulary file: the byte itself if printable, else U+0100 + k */
static int byte_token_spelling(int t, char* out)
{
	const uint32_t cp = t < 188 ? (uint32_t)clip_tokr_token_to_byte(t) : 256u + (uint32_t)(t - 188);
	return utf8_put(out, cp);
}

MLB_API int clip_tokr_load_merges_txt(ClipTokenizer* T, const char* path, int max_merges)
{
	if (!T || !path) return mlsd_set_error(-1, "clip_tokr_load_merges_txt: bad arguments");
	if (max_merges <= 0) max_merges = 49152 - 256 - 2;     /* OpenAI simple_tokenizer.py keeps merges[1:48895] */
	FILE *f = fopen(path, "rb");
	if (!f) return mlsd_set_error(-1, "clip_tokr_load_merges_txt: cannot open '%s'", path);
	SymMap M; memset(&M, 0, sizeof(M));
	uint32_t cap = 1u << 17;
	while (cap < (uint32_t)(512 + max_merges) * 2u) cap <<= 1;
	M.mask = cap - 1; M.off = calloc(cap, sizeof(uint32_t)); M.id = malloc(cap * sizeof(int32_t));
	M.cap = 1 << 20; M.pool = malloc(M.cap); M.used = 1;   /* offset 0 = empty slot */
	int32_t *pairs = malloc(sizeof(int32_t) * 2 * max_merges);
	int n = 0, rc = -1;
	if (!M.off || !M.id || !M.pool || !pairs) { mlsd_set_error(-1, "clip_tokr_load_merges_txt: out of memory"); goto done; }
	for (int t=0;t<256;++t) {
		char s[16]; int l = byte_token_spelling(t, s);
		if (sym_add(&M, s, l, t)) goto done;
		memcpy(s + l, "</w>", 4);
		if (sym_add(&M, s, l + 4, t + 256)) goto done;
	}
	char line[1024];
	int lineno = 0;
	while (n < max_merges && fgets(line, sizeof(line), f)) {
		++lineno;
		size_t L = strlen(line);
		while (L && (line[L-1] == '\n' || line[L-1] == '\r')) line[--L] = 0;
		if (lineno == 1 && line[0] == '#') continue;       /* "#version: 0.2" */
		if (!L) continue;
		char *sp = strchr(line, ' ');
		if (!sp || sp == line || !sp[1] || strchr(sp + 1, ' ')) { mlsd_set_error(-1, "%s:%d: expected two symbols", path, lineno); goto done; }
		const size_t la = sp - line, lb = L - la - 1;
		const int a = sym_find(&M, line, la), b = sym_find(&M, sp + 1, lb);
		if (a < 0 || b < 0) { mlsd_set_error(-1, "%s:%d: merge of an undefined symbol", path, lineno); goto done; }
		char cat[1024];
		memcpy(cat, line, la); memcpy(cat + la, sp + 1, lb);
		if (sym_find(&M, cat, la + lb) >= 0) { mlsd_set_error(-1, "%s:%d: symbol defined twice", path, lineno); goto done; }
		if (sym_add(&M, cat, la + lb, 512 + n)) goto done;
		pairs[2*n] = a; pairs[2*n+1] = b; ++n;
	}
	rc = clip_tokr_set_merges(T, pairs, n);
	if (rc > 0) rc = n;
done:
	fclose(f); free(M.off); free(M.id); free(M.pool); free(pairs);
	return rc;
}

/* ---- tokenisation */
/* next word of [*pc, end): returns its begin, sets *pc to its end; empty word = end of text */
static const char* next_word(const char** pc, const char* end)
{
	const char *cur = *pc;
	while (cur < end) {                               /* skip ASCII and Unicode (category Z) whitespace */
		if (is_ascii_space((unsigned char)*cur)) { ++cur; continue; }
		const char *p = cur;
		if (uc_category(utf8_next(&p, end)) != 'Z') break;
		cur = p;
	}
	const char *beg = cur;
	int run = 0;
	static const char *const contr[] = { "'s", "'t", "'re", "'ve", "'m", "'ll", "'ve", NULL };
	while (cur < end) {
		int matched = 0;
		for (int i=0; contr[i] && !matched; ++i) {
			const char *s = contr[i], *c = cur;
			for (; c < end && *s; ++c, ++s) {
				const int ch = (*c >= 'A' && *c <= 'Z') ? *c + 32 : *c;
				if (ch != *s) break;
			}
			if (!*s) { matched = 1; if (!run) cur = c; }   /* a contraction starts a new word, or is the word */
		}
		if (matched) break;
		const char *p = cur;
		const uint32_t cp = utf8_next(&p, end);
		int cat = is_ascii_space(cp) ? 'Z' : uc_category(cp);
		if (cat == 'Z') break;
		if (cat != 'N' && cat != 'L') cat = 'P';
		if (!run) run = cat;
		else if (cat != run) break;
		cur = p;
	}
	*pc = cur;
	return beg;
}

MLB_API int clip_tokenize(const ClipTokenizer* T, const char* text, int64_t len, int32_t* out, int max_out)
{
	if (!T || !text || !out) return mlsd_set_error(-1, "clip_tokenize: null argument");
	if (len < 0) len = (int64_t)strlen(text);
	const char *cur = text, *end = text + len;
	int pos = 0;
	for (;;) {
		const char *wb = next_word(&cur, end), *we = cur;
		if (we == wb) break;
		/* word -> lower-cased UTF-8 bytes -> byte tokens */
		int32_t *w = out + pos;
		int cnt = 0;
		for (const char *c = wb; c < we; ) {
			char b[4];
			const int nb = utf8_put(b, uc_lower(utf8_next(&c, we)));
			for (int i=0;i<nb;++i) {
				if (pos + cnt >= max_out) return mlsd_set_error(-2, "clip_tokenize: output buffer too small (%d tokens)", max_out);
				w[cnt++] = clip_tokr_byte_to_token(b[i]);
			}
		}
		if (!cnt) continue;
		w[cnt-1] += 256;                                  /* end-of-word mark */
		while (cnt > 1) {                                 /* apply the lowest-ranked adjacent merge */
			int32_t best = INT32_MAX; int at = 0;
			for (int i=1;i<cnt;++i) {
				const int32_t t = merge_rank(T, w[i-1], w[i]);
				if (t < best) { best = t; at = i; }
			}
			if (best == INT32_MAX) break;
			w[at-1] = best;
			memmove(w + at, w + at + 1, sizeof(int32_t) * (size_t)(cnt - at - 1));
			--cnt;
		}
		pos += cnt;
	}
	return pos;
}

/* bytes a token stands for (merges expanded); returns the byte count, sets *end_of_word */
MLB_API int clip_token_decode(const ClipTokenizer* T, int32_t token, char* out, int max, int* end_of_word)
{
	if (!T || token < 0 || token >= 512 + T->n) return mlsd_set_error(-1, "clip_token_decode: token %d is not a byte or merge token", token);
	int32_t stack[64]; int sp = 0, n = 0, eow = 0;
	stack[sp++] = token;
	while (sp) {
		const int32_t t = stack[--sp];
		if (t < 512) {
			if (t >= 256) eow = 1;
			if (n < max) out[n] = (char)clip_tokr_token_to_byte(t);
			++n;
		} else {
			if (sp + 2 > 64) return mlsd_set_error(-1, "clip_token_decode: merge chain too deep");
			stack[sp++] = T->pairs[2*(t-512)+1];
			stack[sp++] = T->pairs[2*(t-512)];
		}
	}
	if (end_of_word) *end_of_word = eow;
	return n;
}
