#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_c6}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json; echo
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/ks.log 2>&1
db=$(find $OUT/ks -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats.txt > /dev/null
rm -rf $OUT/ks
head -8 $OUT/kernel_stats.txt
cd $R
timeout 200 python bench.py --case case141_deep --no-cpu-baseline --no-traffic > $OUT/bench_case141_deep.json 2>> $OUT/bench.err; cut -c1-330 $OUT/bench_case141_deep.json; echo
grep -o '"kernel_avg_ms": [0-9.]*' $OUT/bench.json $OUT/bench_case141_deep.json
