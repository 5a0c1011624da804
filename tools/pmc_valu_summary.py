"""Per-kernel VALU / matrix-pipe occupancy from one rocprofv3 --pmc pass (SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CU_CYCLES SQ_INSTS_VALU) over tools/unet_eval.py: which of the two pipes a kernel keeps busier.

usage: python3 tools/pmc_valu_summary.py <out.json> <counter_collection.csv>
valu_active_frac = SQ_ACTIVE_INST_VALU / (4 SIMDs x SQ_BUSY_CU_CYCLES)   (cycles a SIMD had a VALU instruction executing)
mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)"""
import csv, json, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__file__))
from pmc_summary import label

out, f = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for row in csv.DictReader(open(f, newline="")):
    k = label(row["Kernel_Name"])
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[k].add(row["Dispatch_Id"])
res = {}
for k, d in acc.items():
    busy = d.get("SQ_BUSY_CU_CYCLES", 0.0)
    res[k] = {"launches": len(cnt[k]),
              "valu_active_frac": round(d.get("SQ_ACTIVE_INST_VALU", 0.0) / (4.0 * busy), 4) if busy else None,
              "mfma_busy_frac": round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy), 4) if busy else None,
              "valu_insts_per_launch": round(d.get("SQ_INSTS_VALU", 0.0) / max(len(cnt[k]), 1))}
json.dump({"note": __doc__, "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["launches"])[:12]:
    print(f"{k:36s} n={v['launches']:5d} valu_active={v['valu_active_frac']} mfma_busy={v['mfma_busy_frac']} valu insts/launch={v['valu_insts_per_launch']}")
