/* ORACLE / REFERENCE HARNESS — test infrastructure only.
 * C-string entry point over the REFERENCE's own prompt parser (src/prompt_preproc.h, header-only), compiled by
 * oracle/Makefile with the ccommon sources it needs where they lie under $(REF); nothing of the reference is copied.
 *   int ref_prompt_parse(const char* text, int raw, char* out, int out_size)
 * writes "text\x1f" then per chunk "begin,len,w\x1e" then "\x1f" then per lora "name\x1dw\x1e"; returns the parser's result. */
#include "prompt_preproc.h"
#include <stdio.h>
#include <string.h>

__attribute__((visibility("default")))
int ref_prompt_parse(const char* text, int raw, char* out, int out_size)
{
	PromptText pt = {0};
	int r = 1;
	if (raw) prompt_text_set_raw(&pt, strsl_fromz(text));
	else r = prompt_text_set_parse(&pt, strsl_fromz(text));
	int n = 0;
	out[0] = 0;
	if (r >= 0) {
		n += snprintf(out + n, out_size - n, "%.*s\x1f", (int)dstr_count(pt.text), pt.text ? pt.text : "");
		vec_forp(struct PromptTextChunk, pt.chunks, c, 0)
			n += snprintf(out + n, out_size - n, "%d,%d,%.9g\x1e", (int)(c->text.b - pt.text), (int)c->text.s, c->w);
		n += snprintf(out + n, out_size - n, "\x1f");
		vec_forp(struct PromptTextLora, pt.loras, l, 0)
			n += snprintf(out + n, out_size - n, "%.*s\x1d%.9g\x1e", (int)l->name.s, l->name.b, l->w);
	}
	prompt_text_free(&pt);
	return r;
}
