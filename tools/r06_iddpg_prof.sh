R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06_iddpg; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o ks -- python $R/examples/train_ddpg.py --case case141 --envs 4096 --alg iddpg --episodes 2 --intensity reference > /dev/null 2>> $OUT/prof.err
db=$(find $OUT/prof -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats.txt > /dev/null; rm -rf $OUT/prof
head -24 $OUT/kernel_stats.txt | cut -c1-140
