#!/bin/bash
# HEAD check on the GPU box: GPU tests, bench line, kernel trace (per-kernel stats + launch gaps).  Writes gpurun_out/$TAG/.
TAG=${1:-head}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
timeout 300 python $R/bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/ks.log 2>&1
db=$(find $OUT/ks -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats.txt > /dev/null && python $R/tools/launch_gaps.py $db > $OUT/gaps.txt
rm -rf $OUT/ks
cat $OUT/bench.json | cut -c1-400
cat $OUT/gaps.txt | head -30
