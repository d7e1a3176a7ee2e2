R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_rh_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o c -- python3 $R/tools/prof_reinhard.py f32 > $O/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -o c -- python3 $R/tools/prof_reinhard.py f32 > $O/p2.log 2>&1
python3 - <<PY
import csv, glob, re, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    per=collections.defaultdict(float); names={}; grids={}
    for r in csv.DictReader(open(path)):
        m=re.search(r"sx::reinhard::(\w+)", r["Kernel_Name"])
        if not m: continue
        k=(r["Dispatch_Id"], r["Counter_Name"]); per[k]+=float(r["Counter_Value"]); names[k]=m.group(1); grids[k]=r["Grid_Size"]
    usual={}
    for k in set(names.values()):
        seen=[grids[x] for x in names if names[x]==k]; usual[k]=max(set(seen), key=seen.count)
    for k,v in per.items():
        if grids[k]==usual[names[k]]: acc[k[1]][names[k]].append(v)
for c in sorted(acc):
    print(c.ljust(28), *[f"{k}={sum(v)/len(v):.0f}" for k,v in sorted(acc[c].items())])
PY
