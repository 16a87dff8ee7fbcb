"""Summarise a rocprofv3 --kernel-trace CSV: per-level durations of the last Cholesky factorisation.
usage: python tools/level_profile.py <dir with *_kernel_trace.csv> [last|largest]"""
import csv, glob, sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into factorizations by k_chol_begin
facts, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "k_chol_begin" in n:
        cur = []
        facts.append(cur)
    elif "k_chol_end" in n:
        cur = None
    elif cur is not None and ("k_chol_level" in n or "k_chol_tail" in n):
        cur.append((n, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)), int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 0, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
which = sys.argv[2] if len(sys.argv) > 2 else 'last'
f = max(facts, key=lambda x: sum(y[2] for y in x)) if which == 'largest' else facts[-1]
tot = defaultdict(lambda: [0, 0])
for n, d, gx, wx, s, e in f:
    k = n[n.find("<"):n.find(">") + 1]
    tot[k][0] += d; tot[k][1] += 1
print("levels", len(f), "span_us", (f[-1][5] - f[0][4]) / 1e3, "sum_kernel_us", sum(x[1] for x in f) / 1e3)
for k, v in tot.items():
    print(k, "n", v[1], "total_us", v[0] / 1e3)
for i, (n, d, gx, wx, s, e) in enumerate(f):
    gap = (s - f[i - 1][5]) / 1e3 if i else 0
    print(i, n[n.find("<"):n.find(">") + 1], "wgs", gx // max(wx, 1), "us", d / 1e3, "gap", gap)
