"""Per-kernel totals of a rocprofv3 kernel_stats.csv: name, calls, average us, share."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f'{r["Name"][:64]:64s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s} %')
