#!/usr/bin/env python3
"""Mean per launch of every SQ counter `tools/pmc_sq.sh` collected, one column per transform kernel:
    python3 tools/sq_summary.py gpurun_out/sq > profiles/rNN_sq_counters.csv
(reads <dir>/p*/c_counter_collection.csv; per kernel only the launches with its most frequent grid size count -- bench.py's
timed and warm-up calls, not its one small verification call)."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sq"
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(f"{root}/p*/*counter_collection.csv")):
    per_dispatch = defaultdict(float)
    names, grids = {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r"sx::macenko::(\w+)", r["Kernel_Name"])
            if not m:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[key] = m.group(1)
            grids[key] = r["Grid_Size"]
    usual = {}
    for k in set(names.values()):
        seen = [grids[key] for key in names if names[key] == k]
        usual[k] = max(set(seen), key=seen.count)
    for key, v in per_dispatch.items():
        if grids[key] == usual[names[key]]:
            acc[key[1]][names[key]].append(v)
kernels = sorted({k for c in acc.values() for k in c})
w = csv.writer(sys.stdout)
w.writerow(["Counter_Name", *kernels])
for counter in sorted(acc):
    w.writerow([counter, *[sum(acc[counter][k]) / len(acc[counter][k]) if acc[counter].get(k) else "" for k in kernels]])
