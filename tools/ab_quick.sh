#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  echo "== $lib"
  STAINX_HIP_LIB=$R/stainx_amd/_lib/$lib timeout -k 10 120 python3 $R/tools/bench_twopass.py 
done
