#!/bin/bash
# debug: build timing-experiment variants of the library side by side (mapdn_amd/lib_x<N>.so), selected with MAPDN_LIB_PATH
cd "$(dirname "$0")/.."
for x in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -DMAPDN_NR_STAMPS -DMAPDN_EXP=$x \
      -o mapdn_amd/lib_x$x.so mapdn_amd/csrc/plan.cpp mapdn_amd/csrc/kernels.hip mapdn_amd/csrc/dense.hip mapdn_amd/csrc/sparse.hip mapdn_amd/csrc/policy.hip mapdn_amd/csrc/capi.hip 2>&1 | grep -E "error" ) &
done
wait
ls -la mapdn_amd/lib_x*.so
