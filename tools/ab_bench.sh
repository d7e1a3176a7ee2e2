#!/bin/bash
# A/B of two library builds on bench.py's headline (rotating batches), alternating, plus a kernel trace of each:
#   LIBS="libstainx_prev.so libstainx_hip.so" bash tools/ab_bench.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
LIBS=${LIBS:-"libstainx_prev.so libstainx_hip.so"}
for rep in 1 2; do
  for lib in $LIBS; do
    echo -n "$lib  "
    STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib timeout -k 10 200 python3 $R/bench.py --no-cpu --steps 500 --warmup 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], 'median', r['device_ms_median'], 'min', r['device_ms_min'], 'hot', r['device_ms_hot'])" || exit 1
  done
done
for lib in $LIBS; do
  export STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib
  mkdir -p $R/gpurun_out/ab_bench
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_bench/$lib -o kt -- python3 $R/bench.py --no-cpu --steps 200 --warmup 30 > $R/gpurun_out/ab_bench/$lib.log 2>&1 || exit 1
  echo "== $lib"; python3 $R/tools/profile_summary.py $R/gpurun_out/ab_bench/$lib 100 | cut -c1-60,100-140 | grep -v copyBuffer
done
