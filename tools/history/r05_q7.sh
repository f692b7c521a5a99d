#!/bin/bash
# NOTE: kept as the record of how the experiment was run — the debug library / switch it uses (lib_ab32.so, MAPDN_NR_PAIRS, lib_defer.so) was removed
# again once the result was in profiles/ (f32 mirror: r05_f32_obs_mirror_ab.txt; chain pairs: commit 2ce82a4 + r05_chain_pair_fusion_experiment.txt;
# deferred update: r05_xprop_deferred_update_ab.txt).  It does not run against the current tree.
# round 5, question 7: chain-pair fusion — parity first (solve-only vs oracle, geometry equality, shipped configs), then launch times with / without
mkdir -p gpurun_out; O=gpurun_out/r05_q7.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve_only or nonconvergence or geometry or ieee33 or two_handles or mismatch_pass" 2>&1 | grep -v amdgpu.ids | tail -15 | tee -a $O
timeout 600 python -m pytest tests/test_gpu_shipped_configs.py tests/test_nr_tolerance_edge.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O
for CB in "case141 4096" "case33 4096" "case322 1024" "case141_deep 4096"; do set -- $CB
  for P in 1 0; do
    env MAPDN_NR_PAIRS=$P MAPDN_DEBUG_GEOMETRY=1 timeout 120 python tools/nr_only.py --case $1 --envs $2 --iters 30 2>&1 | grep -v amdgpu.ids | grep "nr kernel\|geometry" | tr '\n' ' ' | sed "s/$/ [pairs $P]\n/" | cut -c1-420 | tee -a $O
  done
done
