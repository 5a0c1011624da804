"""Merge table lines written by tools/tune_all.py / tools/tune_inplan.py into mlimgsynth_amd/csrc/host/tune_table.inc: a line replaces the
line of the same key (the first 12 fields), new keys are appended.  usage: python3 tools/merge_tune.py <new.inc> [<new2.inc> ...]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "mlimgsynth_amd", "csrc", "host", "tune_table.inc")
pat = re.compile(r"\{([^}]*)\}")


def parse(fn):
    out = []
    for line in open(fn):
        m = pat.search(line)
        if m:
            out.append(tuple(int(v) for v in m.group(1).split(",")))
    return out


table = parse(path)
index = {t[:12]: i for i, t in enumerate(table)}
rep = add = 0
for fn in sys.argv[1:]:
    for t in parse(fn):
        if t[:12] in index:
            if table[index[t[:12]]] != t: rep += 1
            table[index[t[:12]]] = t
        else:
            index[t[:12]] = len(table); table.append(t); add += 1
with open(path, "w") as f:
    for t in table:
        f.write("\t{ " + ", ".join(str(v) for v in t) + " },\n")
print(f"{path}: {rep} lines replaced, {add} added, {len(table)} total")
