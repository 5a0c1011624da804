"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs) per kernel.

usage: python3 tools/pmc_summary.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced reads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact
for 16-B-per-lane stores."""
import csv, json, re, sys, collections


def label(name):
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (\d+)", name)
    if m:
        bm, bn, bk, wm, wn, conv, ns = m.groups()
        w16 = "w16" if int(wm) * int(wn) == 16 else ""
        return f"gemm<{bm}x{bn}x{bk}s{ns}{w16},{'conv' if conv == 'true' else 'linear'}>"
    # gemm_pp_kernel<BM, BN, CB0, CB1, RESBATCH, CONV, EPI[, SK[, NPH[, SCH]]]>: labels as mlsd_gemm_variant() prints them
    m = re.search(r"gemm_pp_kernel<(\d+), (\d+), \d+, \d+, (?:true|false), (true|false)(?:, \d+)?(?:, (true|false))?(?:, (\d+))?(?:, (\d+))?>", name)
    if m:
        kind = "ppsk" if m.group(4) == "true" else "pp2" if m.group(5) == "2" else "ppb" if m.group(6) == "1" else "pp"
        epi = re.search(r"gemm_pp_kernel<\d+, \d+, \d+, \d+, (?:true|false), (?:true|false), (\d+)", name)
        if epi and epi.group(1) in ("7", "8"):
            return f"gemm<{m.group(1)}x{m.group(2)}x64{kind},linear+layernorm>"
        return f"gemm<{m.group(1)}x{m.group(2)}x64{kind},{'conv' if m.group(3) == 'true' else 'linear'}>"
    m = re.search(r"gemm_w4_kernel<(\d+), (\d+),", name)
    if m:
        return f"gemm<{m.group(1)}x{m.group(2)}x64w4,linear>"
    m = re.search(r"attn_kernel<(\d+)(?:, (?:true|false))?>", name)
    if m:
        return f"attention<{m.group(1)}>"
    m = re.search(r"attn_tk96_kernel<(\d+)", name)
    if m:
        return f"attention<{m.group(1)},one pass>"
    if "attn64x2_kernel" in name:
        return "attention<64,64 rows/wave>"
    for k in ("gn_stats", "gn_apply", "ln_kernel", "splitk_reduce", "softmax_rows"):
        if k in name:
            return k
    return name[:60]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                c = row.get("Counter_Name"); v = row.get("Counter_Value"); k = row.get("Kernel_Name")
                if c not in ("FETCH_SIZE", "WRITE_SIZE") or k is None:
                    continue
                e = acc[label(k)][c]
                e[0] += 1; e[1] += float(v)
    res = {}
    for k, d in acc.items():
        n = max(d["FETCH_SIZE"][0], d["WRITE_SIZE"][0])
        rd = 2.0 * 1024.0 * d["FETCH_SIZE"][1] / max(d["FETCH_SIZE"][0], 1)
        wr = 1024.0 * d["WRITE_SIZE"][1] / max(d["WRITE_SIZE"][0], 1)
        res[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    with open(out, "w") as fh:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read = 2 x FETCH_SIZE x 1024 (gfx950 "
                           "correction), write = WRITE_SIZE x 1024; averages over all launches of the kernel in the run "
                           "(includes the first evaluation's autotune launches)", "kernels": res}, fh, indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{k:40s} n={v['launches']:6d} rd={v['read_bytes_per_launch']/1e6:9.2f} MB wr={v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
