# kernel trace of the default bench.py workload (headline launches only): bash tools/prof_headline.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-headline_kt}; mkdir -p $O
export STAINX_BENCH_NO_REAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --no-cpu --steps 300 --warmup 50 > $O/log.txt 2>&1
python3 $R/tools/profile_summary.py $O 100 > $O/kernel_stats.txt; cat $O/kernel_stats.txt
