cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pooled_kt; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --workload fit_transform_pooled --steps 60 --warmup 10 --no-cpu > $O/log.txt 2>&1
python3 $R/tools/profile_summary.py $O 100 > $O/kernel_stats.txt; cat $O/kernel_stats.txt
