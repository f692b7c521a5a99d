#!/bin/bash
# round 6: the policy update's trunk as HIP launches (forward_train + backward): parity, then the end-to-end loop with / without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_policy_trunk.py -x -q -m gpu > gpurun_out/r06_q8_tests.txt 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r06_q8_tests.txt
timeout 1200 python -m pytest tests/test_learner.py tests/test_critic_head.py tests/test_rollout_glue.py -x -q -m gpu 2>&1 | tail -2
for F in 1 0; do
  echo "== MAPDN_FUSED_POLICY_TRAIN=$F"
  MAPDN_FUSED_POLICY_TRAIN=$F timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['env_steps_per_s']/1e6,3),'M', d['seconds'], d['phase_seconds'], d['mean_train_policy_loss'], d['mean_train_value_loss'])"
done
