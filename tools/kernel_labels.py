"""rocprofv3 kernel name -> the label mlsd_gemm_variant() / bench.py's kernel tables print for the same launch.

One place for every tool that joins a rocprofv3 CSV (kernel stats, --pmc counter collections) with the plan's own op labels
(tools/pmc_summary.py, tools/pmc_mfma_summary.py, tests/test_profile_labels_cpu.py).  The template arguments are parsed
positionally from the demangled name, so an added or re-typed parameter shows up as a KeyError-free miss that the CPU test
catches (known() returns None), not as a silently mis-keyed JSON (round 5: CONV became an int and the bool-only regex fell
through to name[:60] for every ping-pong launch).

Template parameter lists (csrc/hip):
  gemm_kernel   <BM, BN, BK, WAVES_M, WAVES_N, CONV(bool), NSTAGE, DBG, REG, PAR, ST>          gemm_conv.hip
  gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, CONV(int 0/1/2), EPI, SK=false, NPH=4, SCH=0>     gemm_pp.hpp
  gemm_tt_kernel<EPI>                                                                           gemm_tt.hip
  gemm_w4_kernel<BM, BN, ...>, gemm_skinny_kernel<CONV, ...>, attn_kernel<DH, ...>, attn_tk96_kernel<DH, NW>
"""
import re

PP_EPI = {0: "generic", 1: "f16", 2: "f32", 3: "f32+res", 4: "geglu16", 5: "f32+stats", 6: "f32+res+stats", 7: "f32+ln", 8: "f32+res+ln", 9: "q+cross attention"}
TT_EPI = {1: "f16", 2: "f32", 3: "f32+res", 4: "f32+ln", 5: "f32+res+ln", 6: "f32+ln+chain", 7: "f32+res+ln+chain"}

# every GEMM / attention kernel family of csrc/hip: a name that contains one of these MUST be parsed by its family rule in known()
FAMILIES = ("gemm_pp_kernel", "gemm_tt_kernel", "gemm_w4_kernel", "gemm_skinny_kernel", "gemm_kernel", "attn64x2s_kernel", "attn64x2_kernel", "attn64pp_kernel",
            "attn_tk96_kernel", "attn_q_kernel", "attn_kernel", "conv_smalln_kernel")


def _ident(name):
    """Function identifier of a kernel that is not a tile family (norms, reductions, element-wise passes): its own name is its label."""
    if name.startswith("_Z"):
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name) or re.match(r"_Z(\d+)", name)
        if not m:
            return None
        n, at = int(m.group(1)), m.end()
        return name[at:at + n] if len(name) >= at + n else None
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z_][A-Za-z0-9_]*)(?:<[^(]*>)?(?:\(|$)", name)
    return m.group(1) if m else None


def _targs(name, fn):
    """Template arguments of `fn<...>` in a demangled name, or None.  ('true'/'false' -> bool, digits -> int.)"""
    i = name.find(fn + "<")
    if i < 0:
        return None
    j, depth, out, cur = i + len(fn) + 1, 1, [], ""
    while j < len(name) and depth:
        c = name[j]
        if c == "<": depth += 1
        elif c == ">":
            depth -= 1
            if not depth: break
        if c == "," and depth == 1:
            out.append(cur.strip()); cur = ""
        else:
            cur += c
        j += 1
    out.append(cur.strip())
    conv = []
    for a in out:
        a = re.sub(r"^\([^)]*\)", "", a)           # '(anonymous namespace)::X' / '(Kind)3' casts
        if a in ("true", "false"): conv.append(a == "true")
        elif re.fullmatch(r"-?\d+", a): conv.append(int(a))
        else: conv.append(a)
    return conv


def _mangled_targs(name, fn):
    """The same for an Itanium-mangled name (rocprofv3 leaves some kernels mangled): '<len>fnI L[ib]<v>E ... E'."""
    m = re.search(r"\d+" + re.escape(fn) + r"I((?:L[a-z]\d+E)+)E", name)
    if not m:
        return None
    return [(v == "1") if t == "b" else int(v) for t, v in re.findall(r"L([a-z])(\d+)E", m.group(1))]


def known(name):
    """(label, detail) for a kernel of this library, None for anything the table does not know.
    label = what mlsd_gemm_variant() prints (split-K suffix dropped, as bench.py does); detail = the instantiation's epilogue where one label covers several."""
    for fn in FAMILIES:
        a = _targs(name, fn) or _mangled_targs(name, fn)
        if a is None:
            if fn in ("attn64x2s_kernel", "attn64x2_kernel", "attn64pp_kernel") and fn in name: a = []
            else: continue
        try:
            if fn == "gemm_pp_kernel":
                if len(a) == 6: a = a + [0]            # round 1: <BM, BN, CB0, CB1, RESBATCH, CONV>, one generic epilogue
                bm, bn, _cb0, _cb1, _resb, conv, epi = a[:7]
                sk = a[7] if len(a) > 7 else False
                nph = a[8] if len(a) > 8 else 4
                sch = a[9] if len(a) > 9 else 0
                conv = int(conv)                    # bool (rounds 1-4) or int 0 / 1 / 2 (round 5: 2 = through a nearest-2x upsample)
                kind = "ppsk" if sk else "pp2" if nph == 2 else "ppb" if sch == 1 else "pp"
                what = "linear+layernorm" if epi in (7, 8) else "linear+attention" if epi == 9 else "conv" if conv else "linear"
                return f"gemm<{bm}x{bn}x64{kind},{what}>", PP_EPI.get(epi, str(epi)) + (",upsampled" if conv == 2 else "")
            if fn == "gemm_tt_kernel":
                e = a[0]
                what = "linear+layernorm+linear" if e in (6, 7) else "linear+layernorm" if e in (4, 5) else "linear"
                return f"gemm<128x160x64tt,{what}>", TT_EPI.get(e, str(e))
            if fn == "gemm_w4_kernel":
                return f"gemm<{a[0]}x{a[1]}x64w4,linear>", ""
            if fn == "gemm_skinny_kernel":
                return f"gemm<skinny128x64,{'conv' if a[0] else 'linear'}>", ""
            if fn == "gemm_kernel":
                bm, bn, bk, wm, wn, conv, ns = a[:7]
                reg = a[8] if len(a) > 8 else False
                st = a[10] if len(a) > 10 else False
                w = ("w16" if wm * wn == 16 else "w8") if (bm, bn) == (256, 256) and not (bk == 32 and ns == 3 and wm * wn == 8) else ""
                return f"gemm<{bm}x{bn}x{bk}{'r' if reg else 's'}{ns}{w},{'conv' if conv else 'linear'}>", "stats" if st else ""
            if fn == "attn_kernel":
                return f"attention<{a[0]}>", ""
            if fn == "attn_q_kernel":
                return f"attention<{a[0]},queue>", ""
            if fn == "attn_tk96_kernel":
                return f"attention<{a[0]},one pass>", ""
            if fn == "conv_smalln_kernel":
                return "gemm<conv3x3n16,conv>", f"cin {a[0]}, ring {a[1]}"
            if fn == "attn64x2s_kernel":
                return "attention<64,64 rows/wave pipelined>", ""
            if fn == "attn64x2_kernel":
                return "attention<64,64 rows/wave>", ""
            if fn == "attn64pp_kernel":
                return "attention<64,ping-pong>", ""
        except (ValueError, IndexError, TypeError):
            return None
    if "xattn_pack_vt_kernel" in name:
        return "xattn_pack_vt", ""                  # V^T images of the fused cross attentions: once per context (round 6)
    if "gemm" in name or "attn" in name or "conv_smalln" in name:
        return None                                 # a tile family this table does not know: the CPU test fails on it
    k = _ident(name)
    return (k, "") if k else None


def label(name):
    k = known(name)
    return k[0] if k else "?" + name[:60]
