#!/bin/bash
# round 6: the policy forward with transposed products (k_policy_fwd2) against the rounds-2-5 form: parity tests, micro-benchmark, end-to-end loop
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_learner.py tests/test_policy_trunk.py tests/test_rollout_glue.py -x -q -m gpu 2>&1 | tail -3
for V in 0 1; do
  echo "== MAPDN_POLICY_FWD_V1=$V"
  MAPDN_POLICY_FWD_V1=$V timeout 300 python tools/policy_bench.py 2>/dev/null | grep "fused=1"
  MAPDN_POLICY_FWD_V1=$V timeout 600 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['env_steps_per_s']/1e6,3),'M', d['seconds'], d['phase_seconds'])"
done
