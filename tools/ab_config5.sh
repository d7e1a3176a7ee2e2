R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in libstainx_prev.so libstainx_hip.so libstainx_prev.so libstainx_hip.so; do
  echo "== $lib"
  STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib timeout -k 10 200 python3 $R/bench.py --workload module_config5 --no-cpu --steps 500 --warmup 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['device_ms_min'])" || exit 1
done
