#!/bin/bash
# round 4: config-5 end to end (case322 x 8192 envs, MADDPG, rollout + GPU replay + updates) at the reference's update intensity and at
# the light one, with the per-phase split; kernel table of the reference-intensity run; B = 1 drop-in latency
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_e2e}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 500 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3 --intensity reference --phases --log $OUT/e2e_maddpg_case322_b8192_reference.jsonl > $OUT/e2e_ref.log 2>&1; echo "reference rc=$?"; tail -1 $OUT/e2e_ref.log | cut -c1-900
timeout 200 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3 --intensity light --phases --log $OUT/e2e_maddpg_case322_b8192_light.jsonl > $OUT/e2e_light.log 2>&1; echo "light rc=$?"; tail -1 $OUT/e2e_light.log | cut -c1-900
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o ks -- python $R/examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 1 --intensity reference > /dev/null 2>> $OUT/prof.err
db=$(find $OUT/prof -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/e2e_reference_kernel_stats.txt > /dev/null; rm -rf $OUT/prof
head -16 $OUT/e2e_reference_kernel_stats.txt | cut -c1-150
cd $R
timeout 120 python tools/dropin_latency.py > $OUT/dropin_latency.txt 2>&1; tail -5 $OUT/dropin_latency.txt
