#!/bin/bash
# A/B of library variants under build_ab/: bench line (time, live traffic) per variant and shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_ab}; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  for cfg in ${AB_CFGS:-case141:4096 case322:4096 case141_deep:4096}; do
    c=${cfg%%:*}; b=${cfg##*:}
    tag=${v}_${c}_b$b
    MAPDN_LIB_PATH=$R/build_ab/lib_$v.so timeout 300 python $R/bench.py --case $c --envs $b --no-cpu-baseline --no-other-shapes > $OUT/bench_$tag.json 2>> $OUT/bench.err
    python -c "import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us nr', round(r['kernel_avg_ms']*1e3,1), 'traffic MB', round((r.get('traffic') or 0)/1e6,1))" || tail -3 $OUT/bench.err
  done
done
