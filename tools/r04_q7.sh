#!/bin/bash
# round 4: row barriers only where the schedule has a cross-wave hazard — full GPU suite, then same-box A/B against a build that keeps
# a barrier at every row (-DMAPDN_ALL_ROW_BARRIERS)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_q7}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic"
for rep in 1 2; do
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_allbar.so timeout 200 $B > $OUT/bench_allbar_$rep.json 2>> $OUT/bench.err
  timeout 200 $B > $OUT/bench_new_$rep.json 2>> $OUT/bench.err
done
for cfg in case322:1024 case322:4096 case322:8192 case33:4096 case141:8192 case141_deep:4096; do
  c=${cfg%%:*}; b=${cfg##*:}
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_allbar.so timeout 200 $B --case $c --envs $b > $OUT/bench_allbar_${c}_b$b.json 2>> $OUT/bench.err
  timeout 200 $B --case $c --envs $b > $OUT/bench_new_${c}_b$b.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done | tee $OUT/summary.txt
