#!/bin/bash
# sweep NR launch geometry (waves per workgroup W, envs per workgroup L) with the solve-only driver
CASE=${1:-case141}; ENVS=${2:-4096}
for WL in "1 4" "1 8" "1 16" "1 32" "2 4" "2 8" "2 16" "2 32" "4 8" "4 16" "4 32" "8 16" "8 32"; do set -- $WL
  MAPDN_NR_WAVES=$1 MAPDN_NR_LANES=$2 python tools/nr_only.py --case $CASE --envs $ENVS --iters 20 2>&1 | grep -v amdgpu.ids | tail -1
done
