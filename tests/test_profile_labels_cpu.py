"""The join between rocprofv3's kernel names and the plan's own labels (tools/kernel_labels.py) is what puts `roofline.traffic`
into bench.py's line.  Round 5 broke it silently (a template parameter changed type, the summary fell back to truncated names,
the bench line carried traffic = null): these tests make that failure loud on the CPU."""
import csv
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_labels as KL  # noqa: E402

STATS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats.csv")))


def _round(path):
    m = re.match(r"r(\d+)_", os.path.basename(path))
    return int(m.group(1)) if m else -1


@pytest.mark.parametrize("path", STATS, ids=[os.path.basename(p) for p in STATS])
def test_every_kernel_with_one_percent_of_the_time_has_a_label(path):
    """`Name` column of every committed rocprofv3 --stats summary: no name that carries >= 1 % of the time may fall through."""
    rows = list(csv.DictReader(open(path, newline="")))
    assert rows, path
    for r in rows:
        if float(r["Percentage"]) >= 1.0:
            k = KL.known(r["Name"])
            assert k is not None, f"{os.path.basename(path)}: '{r['Name'][:100]}' ({r['Percentage']} % of the time) has no label"
            assert not k[0].startswith("?")


def test_labels_follow_the_template_parameter_lists():
    anon = "void (anonymous namespace)::"
    cases = {
        anon + "gemm_pp_kernel<256, 256, 2, 2, false, 0, 4, false, 4, 0>((anonymous namespace)::GemmP)": ("gemm<256x256x64pp,linear>", "geglu16"),
        anon + "gemm_pp_kernel<256, 256, 2, 2, false, 0, 1, false, 4, 0>((anonymous namespace)::GemmP)": ("gemm<256x256x64pp,linear>", "f16"),
        anon + "gemm_pp_kernel<128, 320, 3, 2, true, 0, 8, false, 4, 0>((anonymous namespace)::GemmP)": ("gemm<128x320x64pp,linear+layernorm>", "f32+res+ln"),
        anon + "gemm_pp_kernel<128, 320, 3, 2, true, 1, 5, false, 2, 0>((anonymous namespace)::GemmP)": ("gemm<128x320x64pp2,conv>", "f32+stats"),
        anon + "gemm_pp_kernel<256, 256, 2, 2, false, 2, 2, false, 2, 0>((anonymous namespace)::GemmP)": ("gemm<256x256x64pp2,conv>", "f32,upsampled"),
        anon + "gemm_pp_kernel<128, 320, 3, 2, true, 1, 0, true, 4, 0>((anonymous namespace)::GemmP)": ("gemm<128x320x64ppsk,conv>", "generic"),
        anon + "gemm_pp_kernel<128, 320, 3, 2, true, true, 5>((anonymous namespace)::GemmP)": ("gemm<128x320x64pp,conv>", "f32+stats"),     # rounds 2-4: CONV a bool
        anon + "gemm_pp_kernel<256, 256, 2, 2, false, false>((anonymous namespace)::GemmP)": ("gemm<256x256x64pp,linear>", "generic"),       # round 1
        anon + "gemm_tt_kernel<1>((anonymous namespace)::TTP)": ("gemm<128x160x64tt,linear>", "f16"),
        anon + "gemm_tt_kernel<5>((anonymous namespace)::TTP)": ("gemm<128x160x64tt,linear+layernorm>", "f32+res+ln"),
        anon + "gemm_kernel<64, 128, 64, 2, 2, true, 2, 0, false, false, false>((anonymous namespace)::GemmP, (anonymous namespace)::GemmP)": ("gemm<64x128x64s2,conv>", ""),
        anon + "gemm_kernel<256, 256, 64, 4, 4, false, 2, 0, false, false, false>(": ("gemm<256x256x64s2w16,linear>", ""),
        anon + "gemm_skinny_kernel<false, 8, 3, 0>((anonymous namespace)::GemmP)": ("gemm<skinny128x64,linear>", ""),
        anon + "attn_kernel<64, true>((anonymous namespace)::AttnP)": ("attention<64>", ""),
        anon + "attn_tk96_kernel<64, 3>((anonymous namespace)::AttnP, int)": ("attention<64,one pass>", ""),
        anon + "attn64x2_kernel<true>((anonymous namespace)::AttnP)": ("attention<64,64 rows/wave>", ""),
        "_ZN12_GLOBAL__N_116ln_stream_kernelILi3EEEvPKfliifS2_S2_PDF16_Pf": ("ln_stream_kernel", ""),
        "_ZN12_GLOBAL__N_19ln_kernelEPKfliifS1_S1_PDF16_Pf": ("ln_kernel", ""),
        "(anonymous namespace)::gn_apply((anonymous namespace)::GnP)": ("gn_apply", ""),
    }
    for name, want in cases.items():
        assert KL.known(name) == want, name
    # a tile family the table does not know must NOT get a made-up label
    assert KL.known(anon + "gemm_new_kernel<1, 2>((anonymous namespace)::GemmP)") is None
    assert KL.label(anon + "gemm_new_kernel<1, 2>(x)").startswith("?")
    assert KL.known(anon + "gemm_pp_kernel<256, 256, 2>((anonymous namespace)::GemmP)") is None     # too few parameters: a miss, not a guess


@pytest.mark.parametrize("workload,B,dominant", [("sdxl", 4, "gemm<256x256x64pp,linear>"), ("sd15", 1, "gemm<64x128x64s2,conv>")])
@pytest.mark.parametrize("kind", ["traffic", "mfma"])
def test_newest_pmc_summary_is_keyed_by_plan_labels(workload, B, dominant, kind):
    """What bench.py looks up (bench.pmc_entry): the NEWEST committed summary of the plan must know the plan's dominant label, and none of its
    keys may be a raw kernel name.  (The -m gpu contract test checks `dominant` against the label the live plan prints.)"""
    sys.path.insert(0, ROOT)
    import bench
    files = bench.pmc_files(workload, B, kind)
    assert files, f"no profiles/r*_{workload}_b{B}_pmc_{kind}.json"
    newest = files[0]
    assert _round(newest) == max(_round(f) for f in files)
    with open(newest) as fh:
        ks = json.load(fh)["kernels"]
    bad = [k for k in ks if k.startswith("?") or "anonymous namespace" in k or k.startswith("void ") or k.startswith("_Z")]
    assert not bad, f"{os.path.basename(newest)}: keys that are raw kernel names: {bad[:3]}"
    assert dominant in ks, f"{os.path.basename(newest)} does not know '{dominant}'"
    e, f = bench.pmc_entry(workload, B, kind, dominant)
    assert f == newest and e is not None
    if kind == "traffic":
        assert e["hbm_bytes_per_launch"] > 0 and e["launches"] > 0
    else:
        assert 0.0 < e["mfma_busy_frac"] <= 1.0
