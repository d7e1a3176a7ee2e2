"""Print a compact per-kernel table from a rocprofv3 kernel_stats.csv (helper for gpurun sessions)."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = 0.0
    for r in rows:
        if int(r["Calls"]) >= int(sys.argv[2]) if len(sys.argv) > 2 else 1:
            avg = float(r["AverageNs"]) / 1e3
            tot += avg
            print(f"{r['Name'][:86]:86s} {r['Calls']:>5s} avg {avg:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
    print(f"sum of averages: {tot:.1f} us")
