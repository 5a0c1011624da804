/* mlimgsynth-amd: command line front end of libmlimgsynth_amd.so, built on the PUBLIC API only (include/mlis_abi.h).
 * Mirrors the reference's CLI (src/main_mlimgsynth.c): the same commands (:30-41, :676-705) and option names (:42-95, short
 * options :152-168), the progress line (:398-442), TENSOR files ("TENSOR F32 n0 n1 n2 n3\n" + raw floats,
 * src/localtensor.c:196-239).  Every library option travels as text through mlis_option_set_str, so the option grammar is
 * the library's, not this file's.  Images: PNG (8-bit gray / RGB / RGBA, non-interlaced) and binary PNM in, PNG or PNM out
 * (by extension); the infotext is stored in a tEXt chunk "parameters".
 * Additions: --batch-size, --aux-dir, --tokens / --ntokens (comma-separated token ids instead of a prompt, for machines
 * without a CLIP vocabulary file).
 */
#include <ctype.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mlis_abi.h"

static int g_verbose = 1;      /* 0 silent, 1 normal, 2+ verbose */

static void say(int lvl, const char* fmt, ...)
{
	if (g_verbose < lvl) return;
	va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
}
#define FAIL(...) do { fprintf(stderr, "error: " __VA_ARGS__); fputc('\n', stderr); return -1; } while (0)

/* ------------------------------------------------------------------ bytes <-> files */
typedef struct { unsigned char* d; size_t n, cap; } Buf;
static void buf_put(Buf* b, const void* p, size_t n)
{
	if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 64; b->d = (unsigned char*)realloc(b->d, b->cap); }
	memcpy(b->d + b->n, p, n); b->n += n;
}
static int file_read(const char* path, Buf* b)
{
	FILE *f = fopen(path, "rb");
	if (!f) FAIL("could not open '%s'", path);
	unsigned char tmp[65536]; size_t n;
	while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf_put(b, tmp, n);
	fclose(f);
	return 1;
}
static int file_write(const char* path, const void* p, size_t n)
{
	FILE *f = fopen(path, "wb");
	if (!f) FAIL("could not create '%s'", path);
	const int ok = fwrite(p, 1, n, f) == n;
	fclose(f);
	if (!ok) FAIL("short write to '%s'", path);
	return 1;
}

/* ------------------------------------------------------------------ PNG / PNM */
static uint32_t crc_tab[256];
static uint32_t crc32_(uint32_t c, const unsigned char* p, size_t n)
{
	if (!crc_tab[1]) for (uint32_t i=0;i<256;++i) { uint32_t k=i; for (int j=0;j<8;++j) k = (k&1) ? 0xEDB88320u ^ (k>>1) : k>>1; crc_tab[i]=k; }
	c = ~c;
	for (size_t i=0;i<n;++i) c = crc_tab[(c ^ p[i]) & 255] ^ (c >> 8);
	return ~c;
}
static void be32(unsigned char* o, uint32_t v) { o[0]=v>>24; o[1]=v>>16; o[2]=v>>8; o[3]=v; }
static void png_chunk(Buf* b, const char* type, const unsigned char* d, size_t n)
{
	unsigned char h[8]; be32(h, (uint32_t)n); memcpy(h+4, type, 4);
	buf_put(b, h, 8);
	if (n) buf_put(b, d, n);
	uint32_t c = crc32_(0, h+4, 4);
	if (n) { c = ~c; for (size_t i=0;i<n;++i) c = crc_tab[(c ^ d[i]) & 255] ^ (c >> 8); c = ~c; }
	unsigned char t[4]; be32(t, c); buf_put(b, t, 4);
}
/* PNG with stored (uncompressed) deflate blocks: no compressor needed, every decoder reads it */
static int png_write(const char* path, const unsigned char* rgb, unsigned w, unsigned h, unsigned c, const char* text)
{
	Buf raw = {0}, z = {0}, out = {0};
	for (unsigned y=0;y<h;++y) { unsigned char f = 0; buf_put(&raw, &f, 1); buf_put(&raw, rgb + (size_t)y*w*c, (size_t)w*c); }
	unsigned char zh[2] = {0x78, 0x01}; buf_put(&z, zh, 2);
	uint32_t a = 1, b2 = 0;
	for (size_t i=0;i<raw.n;++i) { a = (a + raw.d[i]) % 65521; b2 = (b2 + a) % 65521; }
	for (size_t off=0; off<raw.n || off==0; off+=65535) {
		const size_t n = raw.n - off < 65535 ? raw.n - off : 65535;
		unsigned char bh[5] = { (unsigned char)(off + n >= raw.n), (unsigned char)n, (unsigned char)(n>>8), (unsigned char)~n, (unsigned char)(~n>>8) };
		buf_put(&z, bh, 5); buf_put(&z, raw.d + off, n);
		if (!raw.n) break;
	}
	unsigned char ad[4]; be32(ad, (b2 << 16) | a); buf_put(&z, ad, 4);
	buf_put(&out, "\x89PNG\r\n\x1a\n", 8);
	unsigned char ih[13]; be32(ih, w); be32(ih+4, h); ih[8]=8; ih[9]= c==1 ? 0 : c==3 ? 2 : 6; ih[10]=ih[11]=ih[12]=0;
	png_chunk(&out, "IHDR", ih, 13);
	if (text && *text) {
		Buf t = {0}; buf_put(&t, "parameters", 11); buf_put(&t, text, strlen(text));
		png_chunk(&out, "tEXt", t.d, t.n); free(t.d);
	}
	png_chunk(&out, "IDAT", z.d, z.n);
	png_chunk(&out, "IEND", NULL, 0);
	const int r = file_write(path, out.d, out.n);
	free(raw.d); free(z.d); free(out.d);
	return r;
}

/* inflate (RFC 1951): stored, fixed and dynamic Huffman blocks */
typedef struct { const unsigned char* p; size_t n, pos; uint32_t bits; int nbits; } Bits;
static int getbits(Bits* s, int n)
{
	while (s->nbits < n) { if (s->pos >= s->n) return -1; s->bits |= (uint32_t)s->p[s->pos++] << s->nbits; s->nbits += 8; }
	const int v = (int)(s->bits & ((1u << n) - 1)); s->bits >>= n; s->nbits -= n; return v;
}
typedef struct { short count[16], symbol[288]; } Huff;
static void huff_build(Huff* h, const unsigned char* len, int n)
{
	short offs[16]; memset(h->count, 0, sizeof(h->count));
	for (int i=0;i<n;++i) h->count[len[i]]++;
	h->count[0] = 0; offs[1] = 0;
	for (int i=1;i<15;++i) offs[i+1] = offs[i] + h->count[i];
	for (int i=0;i<n;++i) if (len[i]) h->symbol[offs[len[i]]++] = (short)i;
}
static int huff_decode(Bits* s, const Huff* h)
{
	int code = 0, first = 0, index = 0;
	for (int len=1; len<=15; ++len) {
		const int b = getbits(s, 1); if (b < 0) return -1;
		code |= b;
		const int cnt = h->count[len];
		if (code - cnt < first) return h->symbol[index + (code - first)];
		index += cnt; first += cnt; first <<= 1; code <<= 1;
	}
	return -1;
}
static int inflate_(const unsigned char* src, size_t n, Buf* out)
{
	static const short lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
	static const short lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
	static const short dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
	static const short dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
	static const unsigned char order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
	Bits s = { src, n, 0, 0, 0 };
	int last;
	do {
		last = getbits(&s, 1); const int type = getbits(&s, 2);
		if (last < 0 || type < 0) return -1;
		if (type == 0) {
			s.bits = 0; s.nbits = 0;
			if (s.pos + 4 > s.n) return -1;
			const unsigned len = s.p[s.pos] | (s.p[s.pos+1] << 8); s.pos += 4;
			if (s.pos + len > s.n) return -1;
			buf_put(out, s.p + s.pos, len); s.pos += len;
			continue;
		}
		Huff hl, hd; unsigned char lens[320];
		if (type == 1) {
			for (int i=0;i<288;++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
			huff_build(&hl, lens, 288);
			for (int i=0;i<30;++i) lens[i] = 5;
			huff_build(&hd, lens, 30);
		} else if (type == 2) {
			const int b5 = getbits(&s,5), d5 = getbits(&s,5), c4 = getbits(&s,4);
			if (b5 < 0 || d5 < 0 || c4 < 0) return -1;                  /* (end of input inside the header) */
			const int nl = b5+257, nd = d5+1, nc = c4+4;
			if (nl > 286 || nd > 30) return -1;
			unsigned char cl[19]; memset(cl, 0, sizeof(cl));
			for (int i=0;i<nc;++i) { const int v = getbits(&s,3); if (v < 0) return -1; cl[order[i]] = (unsigned char)v; }
			Huff hc; huff_build(&hc, cl, 19);
			for (int i=0; i<nl+nd; ) {
				int sym = huff_decode(&s, &hc); if (sym < 0) return -1;
				if (sym < 16) { lens[i++] = (unsigned char)sym; continue; }
				int rep, val = 0, xb;
				if (sym == 16) { if (!i) return -1; val = lens[i-1]; xb = getbits(&s,2); rep = 3 + xb; }
				else if (sym == 17) { xb = getbits(&s,3); rep = 3 + xb; }
				else { xb = getbits(&s,7); rep = 11 + xb; }
				if (xb < 0 || i + rep > nl + nd) return -1;
				while (rep--) lens[i++] = (unsigned char)val;
			}
			huff_build(&hl, lens, nl); huff_build(&hd, lens + nl, nd);
		} else return -1;
		for (;;) {
			int sym = huff_decode(&s, &hl); if (sym < 0) return -1;
			if (sym < 256) { unsigned char c = (unsigned char)sym; buf_put(out, &c, 1); continue; }
			if (sym == 256) break;
			sym -= 257; if (sym >= 29) return -1;
			const int lx = lext[sym] ? getbits(&s, lext[sym]) : 0;
			if (lx < 0) return -1;
			const int len = lbase[sym] + lx;
			const int ds = huff_decode(&s, &hd); if (ds < 0 || ds >= 30) return -1;
			const int dx = dext[ds] ? getbits(&s, dext[ds]) : 0;
			if (dx < 0) return -1;                                      /* (-1 at the end of the input used to give dist == 0: a read past the buffer) */
			const size_t dist = (size_t)dbase[ds] + (size_t)dx;
			if (dist < 1 || dist > out->n) return -1;
			for (int i=0;i<len;++i) { unsigned char c = out->d[out->n - dist]; buf_put(out, &c, 1); }
		}
	} while (!last);
	return 1;
}

#define IMG_DIM_MAX 65535      /* like MLIS_OPT_IMAGE_DIM's range */
typedef struct { unsigned char* d; unsigned w, h, c; } Img;
static int paeth(int a, int b, int c) { int p=a+b-c, pa=abs(p-a), pb=abs(p-b), pc=abs(p-c); return (pa<=pb && pa<=pc) ? a : (pb<=pc ? b : c); }
static int img_read(const char* path, Img* im)
{
	Buf f = {0};
	if (file_read(path, &f) < 0) return -1;
	memset(im, 0, sizeof(*im));
	if (f.n > 8 && !memcmp(f.d, "\x89PNG\r\n\x1a\n", 8)) {
		Buf z = {0}, raw = {0}; unsigned ct = 0, bd = 0, il = 0;
		for (size_t p=8; p+12<=f.n; ) {
			const uint32_t len = ((uint32_t)f.d[p]<<24)|(f.d[p+1]<<16)|(f.d[p+2]<<8)|f.d[p+3];
			if (p + 12 + len > f.n) break;
			if (!memcmp(f.d+p+4, "IHDR", 4) && len >= 13) {
				im->w = ((uint32_t)f.d[p+8]<<24)|(f.d[p+9]<<16)|(f.d[p+10]<<8)|f.d[p+11];
				im->h = ((uint32_t)f.d[p+12]<<24)|(f.d[p+13]<<16)|(f.d[p+14]<<8)|f.d[p+15];
				bd = f.d[p+16]; ct = f.d[p+17]; il = f.d[p+20];
			}
			else if (!memcmp(f.d+p+4, "IDAT", 4)) buf_put(&z, f.d+p+8, len);
			p += 12 + len;
		}
		im->c = ct == 0 ? 1 : ct == 2 ? 3 : ct == 6 ? 4 : ct == 4 ? 2 : 0;
		if (bd != 8 || !im->c || il || !im->w || !im->h || z.n < 6) { free(f.d); free(z.d); FAIL("'%s': only 8-bit non-interlaced gray/RGB/RGBA PNG is supported", path); }
		if (im->w > IMG_DIM_MAX || im->h > IMG_DIM_MAX) { free(f.d); free(z.d); FAIL("'%s': image larger than %d x %d", path, IMG_DIM_MAX, IMG_DIM_MAX); }   /* header fields are untrusted: keeps every size product below 2^34 */
		if (inflate_(z.d + 2, z.n - 2, &raw) < 0 || raw.n < (size_t)im->h * ((size_t)im->w * im->c + 1)) { free(f.d); free(z.d); free(raw.d); FAIL("'%s': corrupt PNG data", path); }
		const size_t st = (size_t)im->w * im->c;
		im->d = (unsigned char*)malloc(st * im->h);
		for (unsigned y=0;y<im->h;++y) {
			const unsigned char *in = raw.d + (size_t)y*(st+1), ft = in[0]; in++;
			unsigned char *o = im->d + (size_t)y*st; const unsigned char *up = y ? o - st : NULL;
			for (size_t x=0;x<st;++x) {
				const int a = x >= im->c ? o[x-im->c] : 0, b = up ? up[x] : 0, c = (up && x >= im->c) ? up[x-im->c] : 0;
				o[x] = (unsigned char)(in[x] + (ft==1 ? a : ft==2 ? b : ft==3 ? (a+b)/2 : ft==4 ? paeth(a,b,c) : 0));
			}
		}
		free(z.d); free(raw.d); free(f.d);
		return 1;
	}
	if (f.n > 3 && f.d[0] == 'P' && (f.d[1] == '5' || f.d[1] == '6')) {
		size_t p = 2; unsigned v[3];
		for (int i=0;i<3;++i) {
			for (;;) { while (p < f.n && isspace(f.d[p])) p++; if (p < f.n && f.d[p] == '#') { while (p < f.n && f.d[p] != '\n') p++; } else break; }
			v[i] = 0; while (p < f.n && isdigit(f.d[p])) { if (v[i] > 100000000u) v[i] = 100000000u; v[i] = v[i]*10 + (f.d[p++] - '0'); }   /* (saturating: no wrap on long digit strings) */
		}
		p++;
		im->w = v[0]; im->h = v[1]; im->c = f.d[1] == '6' ? 3 : 1;
		if (im->w > IMG_DIM_MAX || im->h > IMG_DIM_MAX) { free(f.d); FAIL("'%s': image larger than %d x %d", path, IMG_DIM_MAX, IMG_DIM_MAX); }
		const size_t need = (size_t)im->w * im->h * im->c;
		if (v[2] != 255 || !need || p > f.n || need > f.n - p) { free(f.d); FAIL("'%s': unsupported PNM (binary, maxval 255 only)", path); }
		im->d = (unsigned char*)malloc(need); memcpy(im->d, f.d + p, need);
		free(f.d);
		return 1;
	}
	free(f.d);
	FAIL("'%s': not a PNG or binary PNM image", path);
}
static int ends_with(const char* s, const char* e) { const size_t a = strlen(s), b = strlen(e); return a >= b && !strcasecmp(s + a - b, e); }
static int img_write(const char* path, const unsigned char* d, unsigned w, unsigned h, unsigned c, const char* text)
{
	if (ends_with(path, ".ppm") || ends_with(path, ".pgm") || ends_with(path, ".pnm")) {
		Buf b = {0}; char hd[64]; const int n = snprintf(hd, sizeof(hd), "P%c\n%u %u\n255\n", c == 1 ? '5' : '6', w, h);
		buf_put(&b, hd, (size_t)n); buf_put(&b, d, (size_t)w*h*c);
		const int r = file_write(path, b.d, b.n); free(b.d); return r;
	}
	return png_write(path, d, w, h, c, text);
}

/* ------------------------------------------------------------------ tensors (src/localtensor.c:196-239) */
static int tensor_save(const MLIS_Tensor* t, const char* path)
{
	Buf b = {0}; char hd[96];
	const int n = snprintf(hd, sizeof(hd), "TENSOR F32 %d %d %d %d\n", t->n[0], t->n[1], t->n[2], t->n[3]);
	buf_put(&b, hd, (size_t)n); buf_put(&b, t->d, mlis_tensor_count(t) * sizeof(float));
	const int r = file_write(path, b.d, b.n); free(b.d); return r;
}
static int tensor_load(MLIS_Tensor* t, const char* path)
{
	Buf f = {0};
	if (file_read(path, &f) < 0) return -1;
	int s[4] = {1,1,1,1}; size_t p = 11, i = 0;
	if (f.n < 24 || memcmp(f.d, "TENSOR F32 ", 11)) { free(f.d); FAIL("file '%s' is not a valid tensor", path); }
	for (; i<4; ++i) {
		int n = 0; while (p < f.n && isdigit(f.d[p])) { if (n > 100000000) n = 100000000; n = n*10 + (f.d[p++] - '0'); }
		s[i] = n;
		if (p >= f.n) break;
		if (f.d[p] == '\n') { p++; break; }
		if (i == 3 || f.d[p] != ' ') { free(f.d); FAIL("file '%s' is not a valid tensor", path); }
		p++;
	}
	size_t cnt = 1;
	for (int k=0;k<4;++k) { if (s[k] <= 0 || (size_t)s[k] > f.n || cnt > f.n / (size_t)s[k]) { cnt = 0; break; } cnt *= (size_t)s[k]; }   /* overflow-safe: every factor and the product are bounded by the file size */
	if (!cnt || p > f.n || cnt > (f.n - p) / 4) { free(f.d); FAIL("file '%s': truncated tensor", path); }
	mlis_tensor_resize(t, s[0], s[1], s[2], s[3]);
	memcpy(t->d, f.d + p, cnt*4);
	free(f.d);
	return 1;
}
static void tensor_from_img(MLIS_Tensor* t, MLIS_Tensor* alpha, const Img* im)
{	/* ltensor_from_image(_alpha), src/localtensor.c:257-287: planar [w,h,c], v/255 */
	const unsigned nc = (alpha && (im->c == 4 || im->c == 2)) ? im->c - 1 : im->c;
	mlis_tensor_resize(t, (int)im->w, (int)im->h, (int)nc, 1);
	if (alpha && nc != im->c) mlis_tensor_resize(alpha, (int)im->w, (int)im->h, 1, 1);
	for (unsigned y=0;y<im->h;++y) for (unsigned x=0;x<im->w;++x) {
		const unsigned char *px = im->d + ((size_t)y*im->w + x)*im->c;
		for (unsigned c=0;c<nc;++c) t->d[(size_t)im->w*im->h*c + (size_t)im->w*y + x] = px[c] / 255.0f;
		if (alpha && nc != im->c) alpha->d[(size_t)im->w*y + x] = px[nc] / 255.0f;
	}
}
static unsigned char* img_from_tensor(const MLIS_Tensor* t)
{	/* ltensor_to_image: clamp to [0,1], truncate (mlimgsynth.c:123-125) */
	const int w = t->n[0], h = t->n[1], c = t->n[2];
	unsigned char *o = (unsigned char*)malloc((size_t)w*h*c);
	for (int y=0;y<h;++y) for (int x=0;x<w;++x) for (int k=0;k<c;++k) {
		float v = t->d[(size_t)w*h*k + (size_t)w*y + x] * 255.0f;
		o[((size_t)y*w + x)*c + k] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
	}
	return o;
}

/* ------------------------------------------------------------------ options */
static const char k_usage[] =
"Usage: mlimgsynth-amd [COMMAND] [OPTIONS]\n\n"
"Commands: generate | list-backends | vae-encode | vae-decode | vae-test | clip-encode | tokenize | check | convert\n\n"
"Generation:  -p --prompt TEXT   -n --nprompt TEXT   -d --image-dim W,H   -i --input PATH   --imask PATH\n"
"             --ilatent PATH   --ilmask PATH   -o --output PATH   --olatent PATH   --no-prompt-parse BOOL\n"
"             --batch-size N   --tokens IDS   --ntokens IDS\n"
"Models:      -m --model PATH|synth:NAME   --tae PATH   --lora PATH[,MULT]   --lora-dir PATH   -b --backend NAME\n"
"             -t --threads N   --unet-split BOOL   --vae-tile N   --weight-type NAME   --model-type NAME   --aux-dir PATH\n"
"Sampling:    -S --seed N   -s --steps N   --method NAME   --scheduler NAME   --s-noise F   --s-ancestral F\n"
"             --cfg-scale F   --clip-skip N   --f-t-ini F   --f-t-end F\n"
"Output:      -v --verbose   -q --quiet   --silent   --debug   -h --help   -V --version\n";

typedef struct {
	const char *cmd, *in_img, *in_mask, *in_lat, *in_lmask, *out_img, *out_lat, *tokens, *ntokens;
	MLIS_Ctx *ctx;
} Cli;

static const struct { char c; const char* name; } k_short[] = {
	{'h',"help"},{'V',"version"},{'v',"verbose"},{'q',"quiet"},{'b',"backend"},{'m',"model"},{'p',"prompt"},{'n',"nprompt"},
	{'d',"image-dim"},{'s',"steps"},{'S',"seed"},{'t',"threads"},{'i',"input"},{'o',"output"},{0,NULL} };

/* returns 1 if `next` was consumed, 0 if not, 2 to stop (help/version), <0 on error */
static int cli_option(Cli* C, const char* name, const char* next)
{
	if (!strcmp(name, "help")) { fputs(k_usage, stdout); return 2; }
	if (!strcmp(name, "version")) { puts("mlimgsynth-amd (libmlimgsynth_amd, API " MLIS_VERSION_STR ")"); return 2; }
	if (!strcmp(name, "verbose")) { g_verbose++; mlis_option_set(C->ctx, MLIS_OPT_LOG_LEVEL, MLIS_LOGLVL__INCREASE); return 0; }
	if (!strcmp(name, "quiet")) { g_verbose = 1; mlis_option_set(C->ctx, MLIS_OPT_LOG_LEVEL, MLIS_LOGLVL_ERROR); return 0; }
	if (!strcmp(name, "silent")) { g_verbose = 0; mlis_option_set(C->ctx, MLIS_OPT_LOG_LEVEL, MLIS_LOGLVL_NONE); return 0; }
	if (!strcmp(name, "debug")) { g_verbose = 3; mlis_option_set(C->ctx, MLIS_OPT_LOG_LEVEL, MLIS_LOGLVL_DEBUG); return 0; }
	const struct { const char* n; const char** dst; } paths[] = {
		{"input",&C->in_img},{"imask",&C->in_mask},{"ilatent",&C->in_lat},{"ilmask",&C->in_lmask},{"output",&C->out_img},
		{"olatent",&C->out_lat},{"tokens",&C->tokens},{"ntokens",&C->ntokens} };
	for (size_t i=0;i<sizeof(paths)/sizeof(*paths);++i) if (!strcmp(name, paths[i].n)) { *paths[i].dst = next; return 1; }
	if (mlis_option_set_str(C->ctx, name, next) < 0) FAIL("option '--%s %s': %s", name, next, mlis_errstr_get(C->ctx));
	return 1;
}

static int cli_parse(Cli* C, int argc, char** argv)
{
	for (int i=1;i<argc;++i) {
		const char *a = argv[i], *next = i+1 < argc ? argv[i+1] : "";
		int r = 0;
		if (a[0] == '-' && a[1] == '-') r = cli_option(C, a+2, next);
		else if (a[0] == '-' && a[1]) {
			for (int j=1; a[j] && r >= 0 && r != 2; ++j) {
				int k = 0; while (k_short[k].c && k_short[k].c != a[j]) k++;
				if (!k_short[k].c) FAIL("Unknown short option '%c'", a[j]);
				r = cli_option(C, k_short[k].name, next);
			}
		}
		else if (!C->cmd) C->cmd = a;
		else FAIL("Excess of positional arguments");
		if (r < 0) return -1;
		if (r == 2) return 0;
		if (r == 1) i++;
	}
	return 1;
}

static int progress_cb(void* ud, MLIS_Ctx* ctx, const MLIS_Progress* p)
{	/* main_mlimgsynth.c:398-442 */
	(void)ud; (void)ctx;
	if (g_verbose < 1) return 0;
	if (p->stage == MLIS_STAGE_DENOISE) fprintf(stderr, "\r%s %d/%d nfe:%d {%.3fs}%s", mlis_stage_desc(p->stage), p->step, p->step_end, p->nfe, p->step_time,
		p->step == p->step_end ? "\n" : "");
	else if (p->step == 0 || g_verbose > 1) fprintf(stderr, "%s%s", mlis_stage_desc(p->stage), p->step_end > 1 ? "...\n" : "\n");
	return 0;
}

static int set_tokens(Cli* C, const char* list, int negative)
{
	int32_t ids[1024]; int n = 0;
	for (const char *p = list; *p && n < 1024; ) {
		char *e; const long v = strtol(p, &e, 10);
		if (e == p) FAIL("bad token list '%s'", list);
		ids[n++] = (int32_t)v; p = e; while (*p == ',' || *p == ' ') p++;
	}
	if (mlis_amd_prompt_tokens_set(C->ctx, ids, NULL, n, negative) < 0) FAIL("%s", mlis_errstr_get(C->ctx));
	return 1;
}

static char* batch_path(const char* path, int idx, int n)
{
	char *o = (char*)malloc(strlen(path) + 24);
	const char *dot = strrchr(path, '.');
	if (n <= 1) strcpy(o, path);
	else if (dot && !strchr(dot, '/')) sprintf(o, "%.*s-%d%s", (int)(dot - path), path, idx + 1, dot);
	else sprintf(o, "%s-%d", path, idx + 1);
	return o;
}

static int cmd_generate(Cli* C)
{
	int tuf = 0;
	if (C->in_img) {
		Img im; if (img_read(C->in_img, &im) < 0) return -1;
		MLIS_Tensor *ti = mlis_tensor_get(C->ctx, MLIS_TENSOR_IMAGE), *tm = mlis_tensor_get(C->ctx, MLIS_TENSOR_MASK);
		tensor_from_img(ti, tm, &im);
		tuf |= MLIS_TUF_IMAGE; if (im.c == 4 || im.c == 2) tuf |= MLIS_TUF_MASK;
		free(im.d);
	}
	if (C->in_mask) {
		Img im; if (img_read(C->in_mask, &im) < 0) return -1;
		Img g = im; unsigned char *mono = NULL;
		if (im.c != 1) { mono = (unsigned char*)malloc((size_t)im.w*im.h); for (size_t i=0;i<(size_t)im.w*im.h;++i) mono[i] = im.d[i*im.c]; g.d = mono; g.c = 1; }
		tensor_from_img(mlis_tensor_get(C->ctx, MLIS_TENSOR_MASK), NULL, &g);
		tuf |= MLIS_TUF_MASK; free(mono); free(im.d);
	}
	if (C->in_lat) { if (tensor_load(mlis_tensor_get(C->ctx, MLIS_TENSOR_LATENT), C->in_lat) < 0) return -1; tuf |= MLIS_TUF_LATENT; }
	if (C->in_lmask) { if (tensor_load(mlis_tensor_get(C->ctx, MLIS_TENSOR_LMASK), C->in_lmask) < 0) return -1; tuf |= MLIS_TUF_LMASK; }
	if (tuf && mlis_option_set(C->ctx, MLIS_OPT_TENSOR_USE_FLAGS, tuf) < 0) FAIL("%s", mlis_errstr_get(C->ctx));
	if (C->tokens && set_tokens(C, C->tokens, 0) < 0) return -1;
	if (C->ntokens && set_tokens(C, C->ntokens, 1) < 0) return -1;
	if (!C->out_img) C->out_img = "output.png";
	const int n = mlis_generate(C->ctx);
	if (n < 0) FAIL("generate: %s", mlis_errstr_get(C->ctx));
	int n_img = 0;
	for (int i=0; ; ++i) {
		MLIS_Image *im = mlis_image_get(C->ctx, i);
		if (!im || !im->d) break;
		n_img++;
	}
	for (int i=0;i<n_img;++i) {
		MLIS_Image *im = mlis_image_get(C->ctx, i);
		char *path = batch_path(C->out_img, i, n_img);
		const int r = img_write(path, im->d, im->w, im->h, im->c, mlis_infotext_get(C->ctx, i));
		say(1, "Saved %s\n", path);
		free(path);
		if (r < 0) return -1;
	}
	if (C->out_lat && tensor_save(mlis_tensor_get(C->ctx, MLIS_TENSOR_LATENT), C->out_lat) < 0) return -1;
	if (!n_img && !C->out_lat) say(1, "no image produced (--no-decode?)\n");
	return 1;
}

static int cmd_vae(Cli* C, int enc, int dec)
{
	MLIS_Tensor img = {0}, lat = {0};
	int R = -1;
	if (enc) {
		Img im; if (!C->in_img) FAIL("vae-encode needs --input"); if (img_read(C->in_img, &im) < 0) return -1;
		tensor_from_img(&img, NULL, &im); free(im.d);
		if (img.n[2] != 3) { fprintf(stderr, "error: the VAE takes RGB images\n"); goto end; }
		if (mlis_image_encode(C->ctx, &img, &lat, 0) < 0) { fprintf(stderr, "error: %s\n", mlis_errstr_get(C->ctx)); goto end; }
		if (!dec) { if (tensor_save(&lat, C->out_lat ? C->out_lat : (C->out_img ? C->out_img : "latent.tensor")) < 0) goto end; }
	} else {
		if (!C->in_lat) { fprintf(stderr, "error: vae-decode needs --ilatent\n"); goto end; }
		if (tensor_load(&lat, C->in_lat) < 0) goto end;
	}
	if (dec) {
		if (mlis_image_decode(C->ctx, &lat, &img, 0) < 0) { fprintf(stderr, "error: %s\n", mlis_errstr_get(C->ctx)); goto end; }
		unsigned char *px = img_from_tensor(&img);
		const int r = img_write(C->out_img ? C->out_img : "output.png", px, (unsigned)img.n[0], (unsigned)img.n[1], (unsigned)img.n[2], NULL);
		free(px);
		if (r < 0) goto end;
	}
	R = 1;
end:
	mlis_tensor_free(&img); mlis_tensor_free(&lat);
	return R;
}

static int cmd_text(Cli* C, int encode)
{
	const char *prompt = NULL;
	mlis_option_get(C->ctx, MLIS_OPT_PROMPT, &prompt);
	if (!prompt) prompt = "";
	if (!encode) {
		int32_t *tok = NULL;
		const int n = mlis_text_tokenize(C->ctx, prompt, &tok, MLIS_SUBMODEL_CLIP);
		if (n < 0) FAIL("tokenize: %s", mlis_errstr_get(C->ctx));
		for (int i=0;i<n;++i) printf("%d%s", tok[i], i+1<n ? " " : "\n");
		return 1;
	}
	MLIS_Tensor *e = mlis_tensor_get(C->ctx, MLIS_TENSOR_TMP), *f = mlis_tensor_get(C->ctx, (MLIS_TensorId)(MLIS_TENSOR_TMP + 1));
	if (mlis_clip_text_encode(C->ctx, prompt, e, f, MLIS_SUBMODEL_CLIP, 0) < 0) FAIL("clip-encode: %s", mlis_errstr_get(C->ctx));
	printf("embed %dx%dx%dx%d  feat %dx%d\n", e->n[0], e->n[1], e->n[2], e->n[3], f->n[0], f->n[1]);
	if (C->out_lat && tensor_save(e, C->out_lat) < 0) return -1;
	return 1;
}

int main(int argc, char** argv)
{
	Cli C; memset(&C, 0, sizeof(C));
	C.ctx = mlis_ctx_create();
	if (!C.ctx) { fprintf(stderr, "error: could not create the library context\n"); return 1; }
	int R = 1;
	const clock_t t0 = clock();
	mlis_option_set(C.ctx, MLIS_OPT_CALLBACK, progress_cb, NULL);
	const int pr = cli_parse(&C, argc, argv);
	if (pr <= 0) { R = pr < 0; goto end; }
	if (!C.cmd) { fputs(k_usage, stderr); goto end; }
	int r;
	if (!strcmp(C.cmd, "generate")) r = cmd_generate(&C);
	else if (!strcmp(C.cmd, "list-backends")) {
		r = 1;
		for (unsigned i=0; ; ++i) {
			const MLIS_BackendInfo *b = mlis_backend_info_get(C.ctx, i, 0);
			if (!b) break;
			printf("%s\n", b->name);
			for (unsigned d=0; d<b->n_dev; ++d) printf("\t%s '%s' %zu MiB free of %zu\n", b->devs[d].name, b->devs[d].desc, b->devs[d].mem_free >> 20, b->devs[d].mem_total >> 20);
		}
	}
	else if (!strcmp(C.cmd, "vae-encode")) r = cmd_vae(&C, 1, 0);
	else if (!strcmp(C.cmd, "vae-decode")) r = cmd_vae(&C, 0, 1);
	else if (!strcmp(C.cmd, "vae-test")) r = cmd_vae(&C, 1, 1);
	else if (!strcmp(C.cmd, "clip-encode")) r = cmd_text(&C, 1);
	else if (!strcmp(C.cmd, "tokenize")) r = cmd_text(&C, 0);
	else if (!strcmp(C.cmd, "convert")) {          /* image I/O only: --input (PNG / PNM) -> --output (PNG / PNM by extension) */
		Img im; r = -1;
		if (!C.in_img || !C.out_img) fprintf(stderr, "error: convert needs --input and --output\n");
		else if ((r = img_read(C.in_img, &im)) > 0) { r = img_write(C.out_img, im.d, im.w, im.h, im.c, NULL); free(im.d); }
	}
	else if (!strcmp(C.cmd, "check")) { fprintf(stderr, "error: 'check' is not implemented (neither in the reference: main_mlimgsynth.c:605-611)\n"); r = -1; }
	else { fprintf(stderr, "error: Unknown command '%s'\n", C.cmd); r = -1; }
	R = r < 0;
	say(2, "Done {%.3fs cpu}\n", (double)(clock() - t0) / CLOCKS_PER_SEC);
end:
	mlis_ctx_destroy(&C.ctx);
	return R;
}
