#!/bin/bash
# round 6: rollout glue kernels (explore + translate, stats, one-launch replay insertion): parity tests + end-to-end loop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_rollout_glue.py tests/test_replay.py tests/test_learner.py -x -q -m gpu > gpurun_out/r06_q5_tests.txt 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r06_q5_tests.txt
timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases --log gpurun_out/r06_q5_e2e.jsonl > gpurun_out/r06_q5_e2e.txt 2>&1; tail -1 gpurun_out/r06_q5_e2e.txt | cut -c1-800
MAPDN_FUSED_ROLLOUT=0 timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases > gpurun_out/r06_q5_e2e_unfused.txt 2>&1; tail -1 gpurun_out/r06_q5_e2e_unfused.txt | cut -c1-800
