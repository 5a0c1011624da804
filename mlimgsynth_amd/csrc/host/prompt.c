/* Prompt pre-processing — behavioural re-creation of the reference's src/prompt_preproc.h:104-209 (stable-diffusion-webui
 * style emphasis and <lora:NAME[:MULT]> options), pinned by the 10 known-answer cases of src/test_prompt_preproc.c:101-126
 * (tests/golden/reference_kats.json) and, in the build container, live against the reference header itself
 * (oracle/_ref/libprompt_ref.so).
 *   "a (dog) x"      -> chunks "a " 1, "dog" 1.1, " x" 1          "[dog]" -> 1/1.1        "((dog))" -> 1.1^2
 *   "(dog:1.5)"      -> explicit weight (only directly inside ONE '(' )
 *   "\(" "\<" "\n"   -> escapes            "BREAK" -> dropped            "<lora:NAME:0.8>" -> lora list, removed from the text
 */
#include "mlblock_int.h"
#include "mlimgsynth_amd.h"
#include <math.h>

#define E_PARSE (-5)   /* MLIS_E_PROMPT_PARSE */

MLB_API void mlis_prompt_free(MLISPrompt* P)
{
	if (!P) return;
	free(P->text); free(P->chunks); free(P->lora_names); free(P->loras);
	memset(P, 0, sizeof(*P));
}

static int chunk_push(MLISPrompt* P, int begin, float w)
{
	P->chunks = (MLISPromptChunk*)realloc(P->chunks, sizeof(MLISPromptChunk) * (P->n_chunk + 1));
	P->chunks[P->n_chunk].begin = begin; P->chunks[P->n_chunk].len = 0; P->chunks[P->n_chunk].w = w;
	return P->n_chunk++;
}

/* prompt_text_set_raw :50-57: one chunk, weight 1, nothing interpreted */
MLB_API int mlis_prompt_set_raw(MLISPrompt* P, const char* text)
{
	mlis_prompt_free(P);
	P->text = strdup(text ? text : "");
	P->n_text = (int)strlen(P->text);
	chunk_push(P, 0, 1.0f);
	P->chunks[0].len = P->n_text;
	return 1;
}

static int option_parse(MLISPrompt* P, const char* b, const char* e)
{	/* prompt_text_option_parse :59-98 */
	if (e - b >= 5 && !memcmp(b, "lora:", 5)) {
		b += 5;
		const char *sep = b;
		while (sep < e && *sep != ':') sep++;
		float mult = 1;
		if (sep < e && *sep == ':') {   /* optional multiplier */
			char *tail = NULL;
			mult = strtof(sep + 1, &tail);
			if (tail != e) return mlsd_set_error(E_PARSE, "prompt: invalid lora multiplier");
		}
		const int len = (int)(sep - b);
		P->lora_names = (char*)realloc(P->lora_names, P->n_lora_chars + len + 1);
		memcpy(P->lora_names + P->n_lora_chars, b, len);
		P->lora_names[P->n_lora_chars + len] = 0;
		P->loras = (MLISPromptLora*)realloc(P->loras, sizeof(MLISPromptLora) * (P->n_lora + 1));
		P->loras[P->n_lora].name_off = P->n_lora_chars; P->loras[P->n_lora].len = len; P->loras[P->n_lora].w = mult;
		P->n_lora++; P->n_lora_chars += len + 1;
		return 1;
	}
	return mlsd_set_error(E_PARSE, "prompt: unknown option '%.*s'", (int)(e - b), b);
}

/* prompt_text_set_parse :108-209 */
MLB_API int mlis_prompt_set_parse(MLISPrompt* P, const char* src)
{
	mlis_prompt_free(P);
	if (!src) src = "";
	const char *end = src + strlen(src);
	P->text = (char*)malloc(strlen(src) * 2 + 2);
	int nt = 0;
	chunk_push(P, 0, 1.0f);
	int n_paren = 0, n_braket = 0;
	for (const char *cur = src; cur < end; ++cur) {
		if (*cur == '\\') {
			if (cur + 1 < end) { cur++; char c = *cur; if (c == 'n') c = '\n'; P->text[nt++] = c; }
		}
		else if (*cur == '(' || *cur == ')' || *cur == '[' || *cur == ']') {
			switch (*cur) { case '(': n_paren++; break; case ')': n_paren--; break; case '[': n_braket++; break; default: n_braket--; }
			if (n_paren < 0 || n_braket < 0) return mlsd_set_error(E_PARSE, "prompt: unmatched ')' or ']'");
			const float w = pow(1.1, n_paren - n_braket);
			MLISPromptChunk *c = &P->chunks[P->n_chunk - 1];
			if (c->begin == nt) c->w = w;                              /* empty chunk: just re-weight it */
			else { c->len = nt - c->begin; chunk_push(P, nt, w); }
		}
		else if (*cur == ':' && (n_paren > 0 || n_braket > 0)) {
			if (!(n_paren == 1 && n_braket == 0)) return mlsd_set_error(E_PARSE, "prompt: custom emphasis multiplier outside of '()'");
			char *tail = NULL;
			float w = 0;
			if (cur + 1 < end) { cur++; w = strtof(cur, &tail); }
			if (!(tail && tail < end && *tail == ')')) return mlsd_set_error(E_PARSE, "prompt: invalid emphasis with ':'");
			cur = tail - 1;
			P->chunks[P->n_chunk - 1].w = w;
		}
		else if (*cur == '<') {
			const char *e = cur + 1;
			while (e < end && *e != '>') ++e;
			if (e >= end || *e != '>') return mlsd_set_error(E_PARSE, "prompt: '<' not matched with '>'");
			if (option_parse(P, cur + 1, e) < 0) return E_PARSE;
			cur = e;
		}
		else if (*cur == 'B' && cur + 5 < end && !memcmp(cur, "BREAK", 5)) cur += 4;
		else P->text[nt++] = *cur;
	}
	P->text[nt] = 0; P->n_text = nt;
	P->chunks[P->n_chunk - 1].len = nt - P->chunks[P->n_chunk - 1].begin;
	return 1;
}
