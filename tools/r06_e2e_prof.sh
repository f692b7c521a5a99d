#!/bin/bash
# round 6: kernel table of the end-to-end loop (2 episodes: the second has the full 4 update rounds); TAG names the output
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${TAG:-r06_e2e}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o ks -- python $R/examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 2 --intensity reference > /dev/null 2>> $OUT/prof.err
db=$(find $OUT/prof -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/e2e_reference_kernel_stats.txt > /dev/null; rm -rf $OUT/prof
head -45 $OUT/e2e_reference_kernel_stats.txt | cut -c1-150
