"""Generic per-kernel table from rocprofv3 --pmc counter_collection.csv files (one or several passes of the same command).
usage: python3 tools/pmc_table.py <out.txt> <counter_collection.csv> [more.csv ...]
Per kernel (name shortened): launches, average duration, and every counter summed over the kernel's dispatches divided by the
number of dispatches that carried it."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)E", name)
    return (m.group(1) if m else name)[:60]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(dict)
for f in sys.argv[2:]:
    for row in csv.DictReader(open(f, newline="")):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[k][row["Counter_Name"]].add((f, row["Dispatch_Id"]))
        dur[k][(f, row["Dispatch_Id"])] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
counters = sorted({c for k in acc for c in acc[k]})
lines = ["kernel".ljust(60) + " launches   avg_us " + " ".join(c.rjust(22) for c in counters)]
for k in sorted(acc, key=lambda k: -sum(dur[k].values())):
    n = len(dur[k])
    row = k.ljust(60) + f" {n:8d} {sum(dur[k].values()) / n / 1e3:8.1f} "
    row += " ".join((f"{acc[k][c] / max(len(ndisp[k][c]), 1):22.4g}" if c in acc[k] else " " * 22) for c in counters)
    lines.append(row)
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
