"""Summarise the PMC passes of tools/pmc_frontend.sh per frontend kernel: counters per dispatch, HBM bytes (FETCH x2 + WRITE, the gfx950 correction of
MI355X_MICROARCH.md; raw next to it), the shares of a wave's cycles (SQ_WAIT_ANY: parked at s_waitcnt / a barrier, SQ_WAIT_INST_ANY: issue stalls,
SQ_ACTIVE_INST_*: issuing), LDS bank-conflict share.  usage: python tools/pmc_frontend_summary.py <out.json> <pmc dir> [<pmc dir> ...]"""
import csv, glob, json, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for d in dirs:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"].replace("sslam::", "").replace("seg::", "").replace("void ", "")
            n = n[: n.find("(")] if "(" in n else n
            if not (n.startswith("k_") and ("chol" not in n and "linearize" not in n and "lm_" not in n)):
                continue
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n][r["Counter_Name"]] += 1
res = {"source": "rocprofv3 --pmc, one counter set per pass (tools/pmc_frontend.sh); workload tools/frontend_kernels.py: 4 calls of 32 frames x 32 boxes, 3 RANSAC + ICP passes",
       "note": "per dispatch; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950: FETCH_SIZE tallies wide reads at half their bytes), hbm_bytes_raw without the factor; "
               "shares are of SQ_WAVE_CYCLES (quad-cycles summed over the kernel's waves)", "kernels": {}}
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    k = {c: d[c] / max(cnt[n][c], 1) for c in d}
    wc = max(k.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    e = {"dispatches": max(cnt[n].values()), "waves": round(k.get("SQ_WAVES", 0)),
         "hbm_bytes": round((2 * k.get("FETCH_SIZE", 0) + k.get("WRITE_SIZE", 0)) * 1024), "hbm_bytes_raw": round((k.get("FETCH_SIZE", 0) + k.get("WRITE_SIZE", 0)) * 1024),
         "fetch_KB": round(k.get("FETCH_SIZE", 0), 1), "write_KB": round(k.get("WRITE_SIZE", 0), 1),
         "wave_cycles": round(wc), "share_wait_any": round(k.get("SQ_WAIT_ANY", 0) / wc, 3), "share_wait_inst": round(k.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
         "share_active_any": round(k.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "share_active_valu": round(k.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
         "share_active_lds": round(k.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3), "share_active_scalar": round(k.get("SQ_ACTIVE_INST_SCA", 0) / wc, 3),
         "insts_valu": round(k.get("SQ_INSTS_VALU", 0)), "insts_salu": round(k.get("SQ_INSTS_SALU", 0)), "insts_lds": round(k.get("SQ_INSTS_LDS", 0)),
         "insts_vmem_rd": round(k.get("SQ_INSTS_VMEM_RD", 0)), "insts_vmem_wr": round(k.get("SQ_INSTS_VMEM_WR", 0)),
         "lds_bank_conflict_share": round(k.get("SQ_LDS_BANK_CONFLICT", 0) / max(k.get("SQ_LDS_IDX_ACTIVE", 0), 1.0), 3),
         "gui_active_cycles": round(k.get("GRBM_GUI_ACTIVE", 0))}
    res["kernels"][n] = e
json.dump(res, open(out, "w"), indent=1)
for n, e in res["kernels"].items():
    print(f"{n[:34]:34s} n {e['dispatches']:3d} hbm {e['hbm_bytes']/1e6:9.2f} MB wait {e['share_wait_any']:.2f} stall {e['share_wait_inst']:.2f} active {e['share_active_any']:.2f} (valu {e['share_active_valu']:.2f} lds {e['share_active_lds']:.2f}) waves {e['waves']}")
