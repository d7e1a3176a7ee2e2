cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/coded_kt; mkdir -p $O
export STAINX_DIAG=1 AB_ONLY=codes
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/tools/ab_coded.py 100 > $O/log.txt 2>&1
python3 $R/tools/profile_summary.py $O 100 > $O/kernel_stats.txt; cat $O/kernel_stats.txt
