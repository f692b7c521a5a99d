#!/bin/bash
# case322 launch geometry sweep: 16 envs per workgroup (one round at 4096 envs) against the default 8 (two rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_geom322}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { tag=$1; b=$2; shift; shift
  env "$@" timeout 300 python $R/bench.py --case case322 --envs $b --no-cpu-baseline --no-other-shapes --no-traffic > $OUT/bench_$tag.json 2>> $OUT/bench.err
  python -c "import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us nr', round(r['kernel_avg_ms']*1e3,1))" || tail -3 $OUT/bench.err
}
for b in 4096 1024; do
run default_b$b $b MAPDN_X=0
run w4l16_b$b $b MAPDN_NR_LANES=16
run w4l16_lean_b$b $b MAPDN_NR_LANES=16 MAPDN_NR_LEAN=1
run w2l16_lean_b$b $b MAPDN_NR_WAVES=2 MAPDN_NR_LANES=16 MAPDN_NR_LEAN=1
run w8l16_b$b $b MAPDN_NR_WAVES=8 MAPDN_NR_LANES=16
done
