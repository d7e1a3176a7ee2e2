R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_pmc5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o c -- python3 $R/tools/prof_config5.py bf16 > $O/p1.log 2>&1
python3 - <<PY
import csv, glob, re, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    per=collections.defaultdict(float); names={}
    for r in csv.DictReader(open(path)):
        m=re.search(r"sx::macenko::(\w+)<[^,]*, [^,]*, (\w+)", r["Kernel_Name"]) or re.search(r"sx::macenko::(\w+)", r["Kernel_Name"])
        if not m: continue
        name=m.group(1)+("_"+m.group(2) if m.lastindex and m.lastindex>1 else "")
        k=(r["Dispatch_Id"], r["Counter_Name"]); per[k]+=float(r["Counter_Value"]); names[k]=name
    for k,v in per.items(): acc[k[1]][names[k]].append(v)
kern=sorted({k for c in acc.values() for k in c})
print("counter".ljust(26), *[k[:18].rjust(19) for k in kern])
for c in sorted(acc):
    print(c.ljust(26), *[f"{sum(acc[c][k])/len(acc[c][k]):19.0f}" if acc[c].get(k) else " "*19 for k in kern])
PY
