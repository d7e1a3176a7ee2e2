#!/bin/bash
# SQ counter passes over the resident Macenko kernel (separate runs, program directly after `--`) -> gpurun_out/sqres/p*
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sqres; mkdir -p $O
ARGS="${ARGS:-64 512 f32 6}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters_available.txt
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/p1 -o c -- python3 $R/tools/prof_resident.py $ARGS > $O/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -o c -- python3 $R/tools/prof_resident.py $ARGS > $O/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SALU --output-format csv -d $O/p3 -o c -- python3 $R/tools/prof_resident.py $ARGS > $O/p3.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o c -- python3 $R/tools/prof_resident.py $ARGS > $O/kt.log 2>&1
ls $O/*; tail -3 $O/p1.log $O/p3.log
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/p*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        if "resident_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in acc.items():
        v = sorted(d.values()); print(p.split("/")[-2], c, "median per launch", v[len(v)//2], "launches", len(v))
PY
