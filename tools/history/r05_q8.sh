#!/bin/bash
# NOTE: kept as the record of how the experiment was run — the debug library / switch it uses (lib_ab32.so, MAPDN_NR_PAIRS, lib_defer.so) was removed
# again once the result was in profiles/ (f32 mirror: r05_f32_obs_mirror_ab.txt; chain pairs: commit 2ce82a4 + r05_chain_pair_fusion_experiment.txt;
# deferred update: r05_xprop_deferred_update_ab.txt).  It does not run against the current tree.
# round 5: x-propagation backward sweep with the voltage update deferred into the next row's shadow (instead of the update pass) — A/B
mkdir -p gpurun_out; O=gpurun_out/r05_q8.txt; : > $O
L=$PWD/mapdn_amd/lib_defer.so
MAPDN_LIB_PATH=$L timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve_only or nonconvergence or geometry" 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O
for rep in 1 2; do
for CB in "case141 4096" "case33 4096" "case322 1024" "case141_deep 4096"; do set -- $CB
  for V in base defer; do
    if [ $V = defer ]; then export MAPDN_LIB_PATH=$L; else unset MAPDN_LIB_PATH; fi
    timeout 120 python tools/nr_only.py --case $1 --envs $2 --iters 40 2>&1 | grep "nr kernel" | sed "s/$/ [$V]/" | tee -a $O
  done
done
done
