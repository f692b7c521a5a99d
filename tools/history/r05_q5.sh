#!/bin/bash
# round 5, question 5: the >= 8192-envs/GPU layouts — every compiled geometry that was never measured there (W = 2 with 8 / 4 envs per
# workgroup and h LDS-resident; 4 envs x 64 workers), against the shipped defaults.  Solve-only launch time (tools/nr_only.py).
mkdir -p gpurun_out; O=gpurun_out/r05_q5.txt; : > $O
run() { CASE=$1; ENVS=$2; shift 2; env "$@" MAPDN_DEBUG_GEOMETRY=1 timeout 120 python tools/nr_only.py --case $CASE --envs $ENVS --iters 20 2>&1 | grep -v amdgpu.ids | grep "nr kernel\|geometry" | tr '\n' ' ' | sed "s/$/ [$*]\n/" | tee -a $O; }
for CB in "case141 8192" "case322 4096" "case322 8192"; do set -- $CB
  run $1 $2 MAPDN_X=default
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=8 MAPDN_NR_LEAN=1
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=8 MAPDN_NR_LEAN=0
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=8 MAPDN_NR_LEAN=0 MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0 MAPDN_NR_LINE_LDS=0
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=8 MAPDN_NR_LEAN=0 MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0 MAPDN_NR_LINE_LDS=0 MAPDN_NR_G_LDS=0
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=4 MAPDN_NR_LEAN=0 MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0 MAPDN_NR_LINE_LDS=0 MAPDN_NR_G_LDS=0
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=4 MAPDN_NR_LEAN=0 MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0 MAPDN_NR_LINE_LDS=0
  run $1 $2 MAPDN_NR_WAVES=4 MAPDN_NR_LANES=4 MAPDN_NR_LEAN=0
  run $1 $2 MAPDN_NR_WAVES=2 MAPDN_NR_LANES=16 MAPDN_NR_LEAN=1
done
