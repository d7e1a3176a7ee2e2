#!/bin/bash
# Collects the round's evidence for bench.py's headline on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag> <git-head>
# -> gpurun_out/<tag>/: bench.json (default bench.py run), bench_pooled.json, kernel-trace stats (csv + summary), the two PMC
#    passes (FETCH_SIZE, WRITE_SIZE: separate runs, program directly after `--`) and hbm_traffic.json (tools/hbm_traffic.py).
# Copy what is to be judged into profiles/ afterwards (tools/collect_profiles.sh does not write there: gpurun_out/ is scratch).
TAG=${1:-r02}; HEAD=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python3 $R/bench.py --workload fit_transform_pooled --steps 200 --warmup 20 > $O/bench_pooled.json 2> $O/bench_pooled.err
export STAINX_BENCH_NO_REAL=1      # the traces below are the headline's launches only (the default line's real-tile record runs other forms)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu --steps 300 --warmup 50 > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu --steps 20 --warmup 5 > $O/pmc_write.log 2>&1
python3 $R/tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json 6 $HEAD > $O/hbm.log 2>&1
python3 $R/tools/profile_summary.py $O/kt 100 > $O/kernel_stats.txt
cat $O/bench.json; cat $O/kernel_stats.txt; grep total_ $O/hbm_traffic.json
