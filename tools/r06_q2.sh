#!/bin/bash
# round 6: head backward with prefetch + batched dper_n + the fused value loss: parity, micro-benchmark, end-to-end loop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_critic_head.py tests/test_learner.py -x -q -m gpu > gpurun_out/r06_q2_tests.txt 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r06_q2_tests.txt
timeout 600 python tools/head_bench.py > gpurun_out/r06_q2_head_bench.txt 2>&1; cat gpurun_out/r06_q2_head_bench.txt | tail -12
timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases --log gpurun_out/r06_q2_e2e.jsonl > gpurun_out/r06_q2_e2e.txt 2>&1; tail -2 gpurun_out/r06_q2_e2e.txt | cut -c1-800
