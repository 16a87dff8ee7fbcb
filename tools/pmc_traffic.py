"""HBM traffic of a kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately).
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> <algorithmic_bytes> <launches> <substr> [<substr> ...]
Sums the counter (KB) over every dispatch whose kernel name contains one of the substrings and divides by
<launches> (how many 'launches' of the roofline entry the profiled command contained; 0 = one per dispatch of the
first named kernel).  FETCH_SIZE is doubled for
gfx950 as MI355X_MICROARCH.md prescribes (wide coalesced reads are tallied at 64 B per 128-B request)."""
import csv, glob, json, sys
from collections import defaultdict


def collect(d, counter, subs):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = defaultdict(float)
    global first_count
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("sslam::", "")
        n = n[: n.find("(")] if "(" in n else n
        if any(s in n for s in subs):
            per[n] += float(r["Counter_Value"])
            if subs[0] in n:
                first_count += 1
    return per


fetch_dir, write_dir, out, alg, launches = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
subs = sys.argv[6:]
first_count = 0
f = collect(fetch_dir, "FETCH_SIZE", subs)
if launches <= 0:   # one launch per dispatch of the first named kernel
    launches = first_count
w = collect(write_dir, "WRITE_SIZE", subs)
fk, wk = sum(f.values()) / launches, sum(w.values()) / launches
res = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
    "kernels": sorted(set(f) | set(w)),
    "per_kernel_KB": {"FETCH_SIZE": {k: v / launches for k, v in f.items()}, "WRITE_SIZE": {k: v / launches for k, v in w.items()}},
    "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
    "hbm_bytes_raw": (fk + wk) * 1024, "hbm_bytes_fetch_x2": (2 * fk + wk) * 1024,
    "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads; both figures given",
    "algorithmic_bytes": alg, "launches_in_profile": launches,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_raw", "hbm_bytes_fetch_x2", "algorithmic_bytes")}))
