"""Per-kernel table (calls, average / min duration, registers) from a rocprofv3 rocpd database (`*_results.db`), for runs
whose output format was not CSV.   python tools/rocpd_summary.py gpurun_out/prof/x_results.db [min-calls]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.display_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), min(d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.display_name order by min(d.start)").fetchall()
tot = 0.0
for name, calls, avg, mn, vg, sg, lds, _ in rows:
    if calls < min_calls: continue
    name = re.sub(r"^void ", "", name)
    print(f"{name[:90]:90s} {calls:6d} avg {avg/1e3:8.1f} us  min {mn/1e3:8.1f}  vgpr {vg:3d} sgpr {sg:3d} lds {lds}")
    tot += avg / 1e3
print(f"sum of averages: {tot:.1f} us")
