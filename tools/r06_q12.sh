#!/bin/bash
# round 6: replay insertion on a side stream (under the next step's policy forward / power flow): parity + end-to-end A/B
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_rollout_glue.py tests/test_replay.py tests/test_learner.py -x -q -m gpu 2>&1 | tail -2
for A in 1 0; do
  echo "== MAPDN_REPLAY_ASYNC=$A"
  MAPDN_REPLAY_ASYNC=$A timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 4 --phases 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['env_steps_per_s']/1e6,3),'M', d['seconds'], d['phase_seconds'], d['mean_train_value_loss'])"
done
