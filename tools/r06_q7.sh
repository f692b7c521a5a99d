#!/bin/bash
# round 6: the joint-action block of the central critic's first layer through the split-K weight gradient (and, as an A/B, the observation block too); IDDPG after its fix
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_critic_head.py tests/test_learner.py -x -q -m gpu 2>&1 | tail -2
for W in 0 1; do
  echo "== MAPDN_TALL_LINEAR_WIDE=$W"
  MAPDN_TALL_LINEAR_WIDE=$W timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['env_steps_per_s']/1e6,3),'M', d['seconds'], d['phase_seconds'])"
done
timeout 600 python examples/train_ddpg.py --case case141 --envs 4096 --alg iddpg --episodes 3 --phases 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('iddpg', round(d['env_steps_per_s']/1e6,3),'M', d['seconds'], d['phase_seconds'])"
