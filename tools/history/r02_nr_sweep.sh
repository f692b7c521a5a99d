#!/bin/bash
# geometry sweep of the NR kernel with the solve-only driver + short bench runs; args: case envs "W L [extra env]" ...
CASE=${1:-case141}; ENVS=${2:-4096}; shift 2
for cfg in "$@"; do
  set -- $cfg
  env MAPDN_NR_WAVES=$1 MAPDN_NR_LANES=$2 ${3:-X=1} ${4:-Y=1} python tools/nr_only.py --case $CASE --envs $ENVS --iters 30 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/[$3 $4] /"
done
