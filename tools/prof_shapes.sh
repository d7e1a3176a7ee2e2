cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for args in "256 224 224 bf16" "64 512 512 u8"; do
tag=$(echo $args | tr ' ' '_')
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/exp/$tag -o p -- python3 $R/tools/bench_twopass.py $args > $R/gpurun_out/exp_$tag.log 2>&1
echo "== $args"; grep "^{" $R/gpurun_out/exp_$tag.log | cut -c1-160
python3 $R/tools/rocpd_summary.py $R/gpurun_out/exp/$tag/p_results.db 100 | grep -v "prior\|pass_a\|estimate_stage"
done
