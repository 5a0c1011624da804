/* ORACLE / REFERENCE HARNESS — test infrastructure only.
 * Thin C entry point over the REFERENCE's own tensor-name conversion, compiled by oracle/Makefile together with
 * $(REF)/src/tensor_name_conv.c (+ ccommon/alloc.c, alloc_gen.c) where those sources lie; nothing of the reference is
 * copied here.  Exposes plain C strings so that tests can call it through ctypes:
 *     int ref_tnconv_sd(const char* name, char* out, int out_size)   -> tnconv_sd's result, converted name in `out`
 */
#include "tensor_name_conv.h"
#include <string.h>

__attribute__((visibility("default")))
int ref_tnconv_sd(const char* name, char* out, int out_size)
{
	DynStr res = NULL;
	int r = tnconv_sd(strsl_fromz(name), &res);
	size_t n = res ? dstr_count(res) : 0;
	if ((int)n >= out_size) n = out_size - 1;
	if (n) memcpy(out, res, n);
	out[n] = 0;
	dstr_free(res);
	return r;
}
