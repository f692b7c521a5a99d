#!/bin/bash
# round 6: critic-head backward with two wavefronts per SIMD (512 threads, <= 256 registers) against the one-wavefront build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for T in 512 256; do
  echo "== MAPDN_HEAD_BWD_THREADS=$T"
  MAPDN_HEAD_BWD_THREADS=$T timeout 900 python -m pytest tests/test_critic_head.py -x -q -m gpu 2>&1 | tail -2
  MAPDN_HEAD_BWD_THREADS=$T timeout 600 python tools/head_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_q6_head_bench_$T.txt
done
timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases > gpurun_out/r06_q6_e2e.txt 2>&1; tail -1 gpurun_out/r06_q6_e2e.txt | cut -c1-700
