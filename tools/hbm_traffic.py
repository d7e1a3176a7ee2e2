"""HBM bytes per Macenko transform call from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md prescribes), written as the JSON that bench.py reports under roofline.traffic.

    python tools/hbm_traffic.py <fetch_dir> <write_dir> <out.json> [calls_to_average] [git_head]

The JSON also records a hash of the kernel sources (bench.py cites the file only while that hash still matches the tree: there is
no .git on the GPU box) and, if given, the commit it was taken at.

Both counters are in KiB; FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request of a wide coalesced stream).
Only the last `calls_to_average` calls of each kernel are used (warm state), averaged per kernel and summed.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def per_kernel(directory, counter, last):
    files = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)
    vals = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void sx::macenko::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            if "sx::" in name or "at::" in name or "elementwise" in name:   # torch's own kernels
                continue
            vals[name].append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    out = {}
    for k, v in vals.items():
        v.sort()
        tail = [x for _, x in v[-last:]]
        out[k] = sum(tail) / len(tail)
    return out


def main():
    fetch_dir, write_dir, out_path = sys.argv[1:4]
    last = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    git_head = sys.argv[5] if len(sys.argv) > 5 else None
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    from bench import source_hash
    fetch = per_kernel(fetch_dir, "FETCH_SIZE", last)
    write = per_kernel(write_dir, "WRITE_SIZE", last)
    table = {}
    for k in sorted(set(fetch) | set(write)):
        if "float" not in k or "double" in k:      # the fit of the reference tile runs the uint8/other instantiations once; not part of a call
            continue
        table[k] = {"fetch_MB_corrected": round(2 * fetch.get(k, 0.0) * 1024 / 1e6, 1), "write_MB": round(write.get(k, 0.0) * 1024 / 1e6, 1)}
    rd = sum(v["fetch_MB_corrected"] for v in table.values())
    wr = sum(v["write_MB"] for v in table.values())
    doc = {"source_hash": source_hash(), "git_head": git_head, "per_kernel": table, "total_read_MB": round(rd, 1), "total_write_MB": round(wr, 1), "total_bytes": int((rd + wr) * 1e6),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean of the last %d transform calls; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request of a wide coalesced stream); counter values are KiB" % last}
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
