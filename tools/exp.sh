cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest $R/tests/test_twopass_gpu.py -q -m gpu -p no:cacheprovider > $R/gpurun_out/r2_twopass_tests.log 2>&1; tail -3 $R/gpurun_out/r2_twopass_tests.log
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/exp/cur -o p -- python3 $R/tools/bench_twopass.py > $R/gpurun_out/exp_cur.log 2>&1
grep "^{" $R/gpurun_out/exp_cur.log
python3 $R/tools/rocpd_summary.py $R/gpurun_out/exp/cur/p_results.db 100 | grep -E "prior|pass_a|stage_kernel|reconstruct"
