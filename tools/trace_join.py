"""Join a rocprofv3 per-dispatch kernel trace of tools/unet_eval.py with the plan's op list: real kernel durations per SHAPE
(no HIP-event overhead; on this runtime a dispatch's [start, end) runs to the start of the next one, so the figure includes the
dispatch gap, ~4.5 us).  usage: python3 tools/trace_join.py <kernel_trace.csv> <oplist.txt> [--ops]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ops = [l.rstrip("\n").split("\t") for l in open(sys.argv[2])]
# the first op of a UNet plan is the input conversion; evaluations = runs between its occurrences
first = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r["Kernel_Name"]]
ev = rows[first[-2]:first[-1]] if len(first) > 1 else rows[first[-1]:]


def kinds(lab):
    if lab.startswith("gemm"): return ("gemm", "splitk_reduce", "skinny")
    if lab.startswith("attention"): return ("attn",)
    if lab.startswith("groupnorm"): return ("gn_",)
    if lab.startswith("layernorm"): return ("ln_",)
    return (lab.split(" ")[0][:8],)


k = 0
per_op = []
for lab, fl, nb in ops:
    ks = kinds(lab)
    t, names = 0, []
    while k < len(ev):
        nm = ev[k]["Kernel_Name"]
        if not any(s in nm for s in ks): break
        if names:   # an op owns its main kernel and the helper kernels right after it (splitk_reduce; gn_stats/finalize + gn_apply)
            if lab.startswith("gemm") and "_reduce" not in nm: break
            if lab.startswith("groupnorm") and "gn_apply" in names[-1]: break
            if not (lab.startswith("gemm") or lab.startswith("groupnorm")): break
        t += int(ev[k]["End_Timestamp"]) - int(ev[k]["Start_Timestamp"]); names.append(nm); k += 1
    per_op.append((lab, float(fl), float(nb), t / 1e3, len(names)))
tot = sum(p[3] for p in per_op)
print(f"# {len(ops)} ops, {k} of {len(ev)} dispatches joined, {tot / 1e3:.3f} ms")
if "--ops" in sys.argv:
    for p in per_op: print(f"{p[0]:64s} {p[3]:8.1f} us  x{p[4]}")
agg = collections.OrderedDict()
for lab, fl, nb, t, n in per_op:
    e = agg.setdefault(lab, [0, 0.0, 0.0, 0.0, 0]); e[0] += 1; e[1] += t; e[2] += fl; e[3] += nb; e[4] += n
print(f"# {'op':62s} {'n':>3s} {'disp':>4s} {'total_us':>9s} {'us/op':>8s} {'TFLOP/s':>8s} {'GB/s':>8s}")
for lab, (c, t, fl, nb, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{lab:64s} {c:3d} {n:4d} {t:9.1f} {t / c:8.1f} {fl / max(t, 1e-9) / 1e6:8.1f} {nb / max(t, 1e-9) / 1e3:8.1f}")
