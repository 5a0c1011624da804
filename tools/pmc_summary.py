"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs) per kernel.

usage: python3 tools/pmc_summary.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced reads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact
for 16-B-per-lane stores."""
import csv, json, re, sys, collections


sys.path.insert(0, __import__("os").path.dirname(__file__))
from kernel_labels import label, known      # one labeller for every tool (round 6: positional template-argument parsing, CPU-tested)


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    det = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                c = row.get("Counter_Name"); v = row.get("Counter_Value"); k = row.get("Kernel_Name")
                if c not in ("FETCH_SIZE", "WRITE_SIZE") or k is None:
                    continue
                e = acc[label(k)][c]
                e[0] += 1; e[1] += float(v)
                kn = known(k)
                if kn and kn[1]:                    # the same counters per INSTANTIATION where one label covers several epilogues (GEGLU / fp16, +stats, upsampled ...)
                    e = det[f"{kn[0]}[{kn[1]}]"][c]
                    e[0] += 1; e[1] += float(v)
    def fold(table):
        res = {}
        for k, d in table.items():
            n = max(d["FETCH_SIZE"][0], d["WRITE_SIZE"][0])
            rd = 2.0 * 1024.0 * d["FETCH_SIZE"][1] / max(d["FETCH_SIZE"][0], 1)
            wr = 1024.0 * d["WRITE_SIZE"][1] / max(d["WRITE_SIZE"][0], 1)
            res[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                      "hbm_bytes_per_launch": round(rd + wr)}
        return dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))
    res, by_inst = fold(acc), fold(det)
    unknown = sorted(k for k in res if k.startswith("?"))
    if unknown:
        print("pmc_summary: kernel names the labeller does not know (tools/kernel_labels.py):", unknown, file=sys.stderr)
    with open(out, "w") as fh:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read = 2 x FETCH_SIZE x 1024 (gfx950 "
                           "correction), write = WRITE_SIZE x 1024; averages over all launches of the kernel in the run "
                           "(includes the first evaluation's autotune launches); by_instantiation: the same per template instantiation where a label covers several", "kernels": res, "by_instantiation": by_inst}, fh, indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{k:40s} n={v['launches']:6d} rd={v['read_bytes_per_launch']/1e6:9.2f} MB wr={v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
