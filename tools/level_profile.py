"""Summarise a rocprofv3 --kernel-trace CSV: per-launch durations of the last LM step (one Cholesky factorisation,
the backward substitution that follows it and everything else up to the next factorisation).
usage: python tools/level_profile.py <dir with *_kernel_trace.csv> [last|largest]"""
import csv, glob, sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
which = sys.argv[2] if len(sys.argv) > 2 else "last"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("sslam::", "")
    return n[: n.find("(")] if "(" in n else n


recs = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
         int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(int(r.get("Workgroup_Size_X", 1)), 1)) for r in rows]
starts = [i for i, r in enumerate(recs) if "k_chol_begin" in r[0]]
iters = [recs[a:b] for a, b in zip(starts, starts[1:] + [len(recs)])]
it = max(iters, key=lambda x: sum(y[3] for y in x)) if which == "largest" else iters[-2 if len(iters) > 1 else -1]
tot = defaultdict(lambda: [0, 0])
for n, s, e, g in it:
    tot[n][0] += e - s
    tot[n][1] += 1
print("step span_us", (it[-1][2] - it[0][1]) / 1e3, "sum of kernel durations_us", sum(v[0] for v in tot.values()) / 1e3)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:60s} n {v[1]:4d} total_us {v[0] / 1e3:10.1f}")
print("-- launches in order (workgroups, us, gap to previous end)")
prev = None
out = []
for n, s, e, g in it:
    gap = 0 if prev is None else (s - prev) / 1e3
    prev = e
    out.append(f"{n.replace('void ', '')[:28]:28s} wg{g:<7d} {(e - s) / 1e3:8.1f}us gap {gap:5.1f}")
for o in out:
    print("   ", o)
