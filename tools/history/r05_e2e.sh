#!/bin/bash
# round 5: config-5 end to end after the replay views and the split-K weight gradients: 3 episodes with the phase split + a kernel table of one episode
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_e2e; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 300 python examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 3 --intensity reference --phases --log $OUT/e2e_maddpg_case322_b8192_reference.jsonl > $OUT/e2e_ref.log 2>&1; tail -1 $OUT/e2e_ref.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o ks -- python $R/examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 1 --intensity reference > /dev/null 2>> $OUT/prof.err
db=$(find $OUT/prof -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/e2e_reference_kernel_stats.txt > /dev/null; rm -rf $OUT/prof
head -28 $OUT/e2e_reference_kernel_stats.txt | cut -c1-150
