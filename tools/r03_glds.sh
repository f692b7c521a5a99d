#!/bin/bash
# experiment: G factors in LDS instead of the records (case141 x 4096): time and HBM traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_glds}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { tag=$1; shift
  env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-other-shapes > $OUT/bench_$tag.json 2>> $OUT/bench.err
  python -c "import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us nr', round(r['kernel_avg_ms']*1e3,1), 'traffic MB', round((r.get('traffic') or 0)/1e6,1))" || tail -3 $OUT/bench.err
}
run glds_norec MAPDN_NR_G_LDS=1 MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0 MAPDN_NR_LINE_LDS=0
run glds_line MAPDN_NR_G_LDS=1 MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0 MAPDN_NR_LINE_LDS=1
run norec_generic MAPDN_NR_REC_LDS=0 MAPDN_NR_FLAT_LDS=0
run l8_glds MAPDN_NR_LANES=8 MAPDN_NR_G_LDS=1
