"""Prompt pre-processing (emphasis, <lora:>, escapes) of the product against the reference: the known-answer cases of
src/test_prompt_preproc.c:101-126 (stored as data below) and, when oracle/_ref is built, the reference parser itself on a
larger set including error cases."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Chunk(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_int), ("len", ctypes.c_int), ("w", ctypes.c_float)]


class Lora(ctypes.Structure):
    _fields_ = [("name_off", ctypes.c_int), ("len", ctypes.c_int), ("w", ctypes.c_float)]


class Prompt(ctypes.Structure):
    _fields_ = [("text", ctypes.c_char_p), ("n_text", ctypes.c_int), ("chunks", ctypes.POINTER(Chunk)), ("n_chunk", ctypes.c_int),
                ("lora_names", ctypes.POINTER(ctypes.c_char)), ("n_lora_chars", ctypes.c_int), ("loras", ctypes.POINTER(Lora)), ("n_lora", ctypes.c_int)]


def parse(text, raw=False):
    from mlimgsynth_amd import _lib
    L = _lib.lib()
    P = Prompt()
    f = L.mlis_prompt_set_raw if raw else L.mlis_prompt_set_parse
    f.argtypes = [ctypes.POINTER(Prompt), ctypes.c_char_p]
    L.mlis_prompt_free.argtypes = [ctypes.POINTER(Prompt)]
    r = f(ctypes.byref(P), text.encode())
    if r < 0:
        L.mlis_prompt_free(ctypes.byref(P))
        return r, None, None, None
    t = P.text[:P.n_text].decode() if P.n_text else ""
    chunks = [(t.encode()[P.chunks[i].begin:P.chunks[i].begin + P.chunks[i].len].decode(), P.chunks[i].w) for i in range(P.n_chunk)]
    names = ctypes.string_at(P.lora_names, P.n_lora_chars) if P.n_lora_chars else b""
    loras = [(names[P.loras[i].name_off:P.loras[i].name_off + P.loras[i].len].decode(), P.loras[i].w) for i in range(P.n_lora)]
    L.mlis_prompt_free(ctypes.byref(P))
    return r, t, chunks, loras


F = ctypes.c_float
KATS = [   # (text, chunks, loras): src/test_prompt_preproc.c:104-123; weights are C float expressions
    ("a dog jumping", [("a dog jumping", 1.0)], []),
    ("a (dog) jumping", [("a ", 1.0), ("dog", 1.1), (" jumping", 1.0)], []),
    ("a [dog] jumping", [("a ", 1.0), ("dog", 1 / 1.1), (" jumping", 1.0)], []),
    ("a ((dog)) jumping", [("a ", 1.0), ("dog", 1.1 * 1.1), (" jumping", 1.0)], []),
    ("a (dog:1.5) jumping", [("a ", 1.0), ("dog", 1.5), (" jumping", 1.0)], []),
    ("a dog jum<lora:LORA NAME>ping", [("a dog jumping", 1.0)], [("LORA NAME", 1.0)]),
    ("a dog jum<lora:LORA NAME:0.8>ping", [("a dog jumping", 1.0)], [("LORA NAME", 0.8)]),
    ("a \\(dog\\) jumping", [("a (dog) jumping", 1.0)], []),
    ("a dog jum\\<lora:LORA NAME>ping", [("a dog jum<lora:LORA NAME>ping", 1.0)], []),
]


def test_reference_known_answers():
    for text, chunks, loras in KATS:
        r, t, c, l = parse(text)
        assert r == 1, text
        assert [x[0] for x in c] == [x[0] for x in chunks], text
        assert [F(x[1]).value for x in c] == [F(x[1]).value for x in chunks], text        # a.w != b.w comparison in float
        assert [(n, F(w).value) for n, w in l] == [(n, F(w).value) for n, w in loras], text
    raw = "a (dog:1.5) jumping [in] the ((park))"
    r, t, c, l = parse(raw, raw=True)
    assert c == [(raw, 1.0)] and l == []


def test_live_against_reference_parser():
    so = os.path.join(ROOT, "oracle", "_ref", "libprompt_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libprompt_ref.so not built (the reference is only present in the build container)")
    ref = ctypes.CDLL(so)
    ref.ref_prompt_parse.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    cases = [k[0] for k in KATS] + [
        "", "()", "(a)(b)[c]", "a ((b) c) d", "x (y:0.25)z", "[[deep]] (mix [of] both)", "line\\nbreak", "a BREAK b BREAKING", "BREAK",
        "<lora:one><lora:two:1.5> tail", "(a:1.2", "a) b", "a] b", "(a [b:2])", "[a:2]", "<lora:bad:1.x>", "<lorax:foo>", "<unclosed", "(w:)",
        "((a:2))", "trailing\\", "(:1.5)", "a (b:1.5e-1) c", "éè (中文:1.3)"]
    buf = ctypes.create_string_buffer(8192)
    for text in cases:
        rr = ref.ref_prompt_parse(text.encode(), 0, buf, 8192)
        r, t, c, l = parse(text)
        assert (r < 0) == (rr < 0), (text, r, rr)
        if rr < 0:
            continue
        parts = buf.value.split(b"\x1f")
        rt = parts[0].decode()
        rc = [tuple(x.split(b",")) for x in parts[1].split(b"\x1e") if x]
        rl = [tuple(x.split(b"\x1d")) for x in parts[2].split(b"\x1e") if x]
        assert t == rt, text
        tb = t.encode()
        assert c == [(tb[int(b):int(b) + int(n)].decode(), F(float(w)).value) for b, n, w in rc], text
        assert l == [(n.decode(), F(float(w)).value) for n, w in rl], text
