#!/bin/bash
# The diagnostic build (what __graft_entry__.build() also makes): -DSX_DIAG -DSX_STAMPS -> stainx_amd/_lib/libstainx_diag.so.
# Tools that force forms / read stage stamps run with STAINX_DIAG=1 (stainx_amd/_native.py), tests ask for it with MacenkoHIP(dev, diag=True).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -DSX_DIAG -DSX_STAMPS ${EXTRA_FLAGS} \
  $R/stainx_amd/csrc/api.hip $R/stainx_amd/csrc/macenko.hip $R/stainx_amd/csrc/reinhard.hip $R/stainx_amd/csrc/histmatch.hip -o $R/stainx_amd/_lib/${OUT:-libstainx_diag.so}
