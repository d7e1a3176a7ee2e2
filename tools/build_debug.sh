#!/bin/bash
# Diagnostic build of the library with the time stamps compiled in (-DSX_STAMPS): stainx_amd/_lib/libstainx_dbg.so, used through
# STAINX_HIP_LIB by tools/bench_twopass.py, tools/stage_stamps.py and tools/fused_timeline.py.  The product build has no stamps.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -DSX_STAMPS ${EXTRA_FLAGS} \
  $R/stainx_amd/csrc/api.hip $R/stainx_amd/csrc/macenko.hip $R/stainx_amd/csrc/reinhard.hip $R/stainx_amd/csrc/histmatch.hip -o $R/stainx_amd/_lib/${OUT:-libstainx_dbg.so}
