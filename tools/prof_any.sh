# kernel-trace summary of any driver script: bash tools/prof_any.sh <tag> <script and args ...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/tools/$@ > $O/log.txt 2>&1
python3 $R/tools/profile_summary.py $O 100 > $O/kernel_stats.txt; cat $O/kernel_stats.txt
