#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03_c14
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 300 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3 --intensity reference --log $OUT/e2e_maddpg_case322_b8192_reference.jsonl > $OUT/e2e_ref.log 2>&1; echo "reference rc=$?"; tail -2 $OUT/e2e_ref.log | cut -c1-330
MAPDN_FUSED_LN=0 timeout 300 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 2 --intensity reference --log $OUT/e2e_maddpg_case322_b8192_reference_torch_ln.jsonl > $OUT/e2e_ref_torchln.log 2>&1; tail -1 $OUT/e2e_ref_torchln.log | cut -c1-330
timeout 200 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3 --intensity light --log $OUT/e2e_maddpg_case322_b8192_light.jsonl > $OUT/e2e_light.log 2>&1; tail -1 $OUT/e2e_light.log | cut -c1-330
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python $R/examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 1 --max-steps 120 --intensity reference > $OUT/e2e_prof.log 2>&1
db=$(find $OUT/ks -name "*.db" | head -1); python $R/tools/prof_summary.py $db $OUT/e2e_reference_kernel_stats.txt | head -14 | cut -c1-140; rm -rf $OUT/ks
cd $R
timeout 400 python examples/learning_curve.py --case case33 --alg iddpg --envs 256 --episodes 300 --out $OUT/curve > $OUT/curve.log 2>&1; tail -1 $OUT/curve.log
