#!/bin/bash
# A/B of two builds of the library on the GPU box: bench_twopass with each, then a kernel trace of the variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libstainx_hip.so libstainx_dbg.so libstainx_dbg2.so}; do
  [ -f $R/stainx_amd/_lib/$lib ] || continue
  echo "== $lib"
  STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib timeout -k 10 120 python3 $R/tools/bench_twopass.py | cut -c1-330
  export STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$lib -o kt -- python3 $R/bench.py --no-cpu --steps 200 --warmup 30 > $R/gpurun_out/ab_$lib.log 2>&1
  python3 $R/tools/profile_summary.py $R/gpurun_out/ab_$lib 100 | cut -c1-60,100-140 | grep -v copyBuffer
done
