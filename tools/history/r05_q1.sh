#!/bin/bash
# round 5, question 1: 4 envs per workgroup (L = 4) — bit-equality, then launch-time sweeps at sub-saturating batches
mkdir -p gpurun_out; O=gpurun_out/r05_q1.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_nr_launch_geometry" 2>&1 | tail -5 | tee -a $O
sweep() { CASE=$1; ENVS=$2; shift 2
  for WL in "$@"; do set -- $WL
    env MAPDN_NR_WAVES=$1 MAPDN_NR_LANES=$2 MAPDN_NR_LEAN=0 $3 $4 $5 timeout 120 python tools/nr_only.py --case $CASE --envs $ENVS --iters 30 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/$/ $3 $4 $5/" | tee -a $O
  done; }
sweep case322 1024 "4 8" "4 4" "4 4 MAPDN_NR_REC_LDS=0 MAPDN_NR_G_LDS=1" "2 4" "2 4 MAPDN_NR_REC_LDS=0 MAPDN_NR_G_LDS=1" "1 4" "2 8"
sweep case141 1024 "4 16" "4 8" "4 4" "2 4" "1 4"
sweep case141 2048 "4 16" "4 8" "4 4"
sweep case141 4096 "4 16" "4 8"
sweep case33 1024 "1 16" "1 8" "1 4" "2 4" "4 4"
sweep case33 4096 "1 16" "1 8" "2 8"
sweep case322 16 "4 8" "4 4" "2 4"
sweep case141 16 "4 16" "4 4" "2 4"
