R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in libstainx_prev.so libstainx_hip.so libstainx_prev.so libstainx_hip.so; do
  echo "== $lib"
  STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib timeout -k 10 120 python3 $R/tools/bench_reinhard.py 2>/dev/null || exit 1
done
for lib in libstainx_prev.so libstainx_hip.so; do
  export STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_rh/$lib -o kt -- python3 $R/tools/prof_reinhard.py f32 > $R/gpurun_out/r03_rh/$lib.log 2>&1 || exit 1
  python3 $R/tools/profile_summary.py $R/gpurun_out/r03_rh/$lib 20
done
