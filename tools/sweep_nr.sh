#!/bin/bash
# sweep NR launch geometry (waves per workgroup W, envs per workgroup L) with the solve-only driver
CASE=${1:-case141}; ENVS=${2:-4096}
for W in 1 2 4; do for L in 64 32 16 8 4; do
  MAPDN_NR_WAVES=$W MAPDN_NR_LANES=$L python tools/nr_only.py --case $CASE --envs $ENVS --iters 20 2>&1 | grep -v amdgpu.ids | tail -1
done; done
