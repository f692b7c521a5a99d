#!/bin/bash
# round-3 GPU call: learning-curve run (case33 IDDPG), config-5 end-to-end at two update intensities, B = 1 drop-in latency
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_e2e}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 400 python examples/learning_curve.py --case case33 --alg iddpg --envs 256 --episodes ${EPISODES:-300} --out $OUT/curve > $OUT/curve.log 2>&1; echo "curve rc=$?"; tail -3 $OUT/curve.log
timeout 200 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3 --intensity light --log $OUT/e2e_maddpg_case322_b8192_light.jsonl > $OUT/e2e_light.log 2>&1; echo "light rc=$?"; tail -2 $OUT/e2e_light.log | cut -c1-400
timeout 500 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 2 --intensity reference --log $OUT/e2e_maddpg_case322_b8192_reference.jsonl > $OUT/e2e_ref.log 2>&1; echo "reference rc=$?"; tail -2 $OUT/e2e_ref.log | cut -c1-400
timeout 120 python tools/dropin_latency.py > $OUT/dropin_latency.txt 2>&1; tail -5 $OUT/dropin_latency.txt
