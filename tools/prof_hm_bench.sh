cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hmk; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --workload hm_config3 --no-cpu --steps 200 --warmup 20 > $O/log.txt 2>&1
python3 $R/tools/profile_summary.py $O 100; tail -1 $O/log.txt | cut -c1-250
