#!/bin/bash
# round-3 check on the GPU box: full GPU suite, then bench line + kernel table for the three shapes at 4096 envs
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_c10}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
for cfg in case141:4096 case322:4096 case33:4096; do
  c=${cfg%%:*}; b=${cfg##*:}
  timeout 200 python $R/bench.py --case $c --envs $b --no-cpu-baseline --no-traffic > $OUT/bench_${c}_b$b.json 2>> $OUT/bench.err
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python $R/bench.py --case $c --envs $b --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/ks.log 2>&1
  db=$(find $OUT/ks -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_${c}_b$b.txt > /dev/null
  rm -rf $OUT/ks
  python -c "import json; d=json.loads(open('$OUT/bench_${c}_b$b.json').read().strip().splitlines()[-1]); print('$c $b', round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us nr', round(d['roofline']['kernel_avg_ms']*1e3,1))"
  head -7 $OUT/kernel_stats_${c}_b$b.txt | tail -5 | cut -c1-125
done
