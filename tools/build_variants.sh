#!/bin/bash
# debug: build timing-experiment variants of the library side by side (mapdn_amd/lib_x<N>.so), selected with MAPDN_LIB_PATH
cd "$(dirname "$0")/.."
for x in "$@"; do
  MAPDN_BUILD_OUT=$PWD/mapdn_amd/lib_x$x.so MAPDN_EXTRA_FLAGS="-DMAPDN_NR_STAMPS -DMAPDN_EXP=$x" python -m mapdn_amd.build --force
done
ls -la mapdn_amd/lib_x*.so
