"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_F16) over tools/unet_eval.py.

usage: python3 tools/pmc_mfma_summary.py <out.json> <counter_collection.csv>
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the busy CU-cycles in which a SIMD's
matrix pipe was executing (gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived metrics).
f16_tflops_from_mops = SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 FLOP / kernel duration (a MOPS unit = 512 FLOP)."""
import csv, json, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__file__))
from kernel_labels import label

out, f = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
cnt = collections.defaultdict(int)
seen = set()
for row in csv.DictReader(open(f, newline="")):
    k = label(row["Kernel_Name"])
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"],)
    if key not in seen:
        seen.add(key)
        dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        cnt[k] += 1
res = {}
for k, d in acc.items():
    busy = d.get("SQ_BUSY_CU_CYCLES", 0.0)
    res[k] = {"launches": cnt[k],
              "mfma_busy_frac": round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy), 4) if busy else None,
              "f16_tflops_from_mops": round(d.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512.0 / max(dur[k], 1.0) / 1e3, 1),
              "avg_us_under_pmc": round(dur[k] / max(cnt[k], 1) / 1e3, 2)}
res = dict(sorted(res.items(), key=lambda kv: -(kv[1]["avg_us_under_pmc"] * kv[1]["launches"])))
json.dump({"note": __doc__, "kernels": res}, open(out, "w"), indent=1)
for k, v in list(res.items())[:10]:
    print(f"{k:36s} n={v['launches']:5d} mfma_busy={v['mfma_busy_frac']}  f16 TFLOP/s(mops)={v['f16_tflops_from_mops']}  avg {v['avg_us_under_pmc']} us")
