#!/bin/bash
# round 6: does the learner still learn through the round-6 kernels?  seeded IDDPG and MADDPG runs on case33 x 256 envs (the round-3 recipe),
# checkpoints evaluated on the CPU ORACLE env
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06_curve; mkdir -p $OUT
for ALG in iddpg maddpg; do
  timeout 900 python examples/learning_curve.py --case case33 --alg $ALG --envs 256 --episodes 300 --ckpt-every 100 --out $OUT/$ALG > $OUT/$ALG.log 2>&1; echo "$ALG rc=$?"; tail -2 $OUT/$ALG.log | cut -c1-300
  timeout 1500 python tools/eval_checkpoints_on_oracle.py $OUT/$ALG --episodes 8 --procs 16 --out $OUT/r06_learning_curve_case33_${ALG}_b256.json > $OUT/eval_$ALG.log 2>&1; echo "eval rc=$?"; tail -8 $OUT/eval_$ALG.log | cut -c1-200
  cp $OUT/$ALG/train.jsonl $OUT/r06_learning_curve_case33_${ALG}_b256_train.jsonl 2>/dev/null
  rm -f $OUT/$ALG/*.pt
done
