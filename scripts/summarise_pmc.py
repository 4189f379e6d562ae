"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter)
-> JSON.  Usage: summarise_pmc.py <counter_collection.csv> <out.json> [kernel-name substring]"""
import csv, json, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
only = sys.argv[3] if len(sys.argv) > 3 else ""
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
dur = defaultdict(float)
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"]
    if only and only not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in calls[k]:
        calls[k].add(r["Dispatch_Id"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, c in acc.items():
    n = len(calls[k])
    e = {"dispatches": n, "mean_duration_ns_under_pmc": dur[k] / n}
    e.update({name: v / n for name, v in sorted(c.items())})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
        e["mfma_busy_over_sq_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if "GRBM_GUI_ACTIVE" in c and dur[k]:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        e["clock_GHz_under_load"] = c["GRBM_GUI_ACTIVE"] / 8 / dur[k]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_over_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    out[k] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
