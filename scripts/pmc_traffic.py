"""FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two separate rocprofv3 --pmc runs of bench.py)
-> HBM bytes per launch and per pair of the similarity kernel.
Usage: pmc_traffic.py <fetch.csv> <write.csv> <kernel substring> <pairs per launch> <out.json>"""
import csv, json, sys

fetch_csv, write_csv, kern, ppl, dst = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]


def mean_per_dispatch(path, counter):
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    return tot / max(len(ids), 1), len(ids)


f, nf = mean_per_dispatch(fetch_csv, "FETCH_SIZE")
w, nw = mean_per_dispatch(write_csv, "WRITE_SIZE")
hbm = 2.0 * f * 1024 + w * 1024
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 0 "
              "--no-cpu-baseline --no-ransac`",
    "kernel": kern, "launches": nf,
    "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
    "correction": "gfx950 FETCH_SIZE tallies 128-B requests of wide (16 B/lane) coalesced reads at 64 B: read bytes = "
                  "2 x FETCH_SIZE; WRITE_SIZE is exact",
    "hbm_bytes_per_launch": hbm, "pairs_per_launch": ppl, "hbm_bytes_per_pair": hbm / ppl,
    "algorithmic_bytes_per_pair_fp32_inputs": 8650752,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
