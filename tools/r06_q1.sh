#!/bin/bash
# round 6, first contact of the one-launch critic head: parity tests, micro-benchmark, the end-to-end loop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_critic_head.py -x -q -m gpu > gpurun_out/r06_q1_tests.txt 2>&1; echo "head tests rc=$?" 
tail -15 gpurun_out/r06_q1_tests.txt
timeout 600 python tools/head_bench.py > gpurun_out/r06_q1_head_bench.txt 2>&1; cat gpurun_out/r06_q1_head_bench.txt | tail -12
timeout 900 python -m pytest tests/test_learner.py -x -q -m gpu > gpurun_out/r06_q1_learner_tests.txt 2>&1; echo "learner tests rc=$?"; tail -5 gpurun_out/r06_q1_learner_tests.txt
timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases --log gpurun_out/r06_q1_e2e.jsonl > gpurun_out/r06_q1_e2e.txt 2>&1; tail -3 gpurun_out/r06_q1_e2e.txt | cut -c1-900
