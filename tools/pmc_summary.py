"""Sum rocprofv3 --pmc counters per kernel name.  usage: python tools/pmc_summary.py <dir>"""
import csv, glob, sys
from collections import defaultdict
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].replace("sslam::", "")
    n = n[: n.find("(")] if "(" in n else n
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(n, r["Counter_Name"])] += 1
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(n, {k: f"{v:.3g}" for k, v in d.items()}, "dispatches", max(cnt[(n, k)] for k in d))
