#!/bin/bash
# SQ counter passes over bench.py (separate runs, program directly after `--`) -> gpurun_out/sq/p{1,2,3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export STAINX_BENCH_NO_REAL=1      # the headline's launches only
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $O/p1 -o c -- python3 $R/bench.py --no-cpu --steps 10 --warmup 3 > $O/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p2 -o c -- python3 $R/bench.py --no-cpu --steps 10 --warmup 3 > $O/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL --output-format csv -d $O/p3 -o c -- python3 $R/bench.py --no-cpu --steps 10 --warmup 3 > $O/p3.log 2>&1
ls $O/*; tail -3 $O/p1.log
