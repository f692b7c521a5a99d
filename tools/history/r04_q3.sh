#!/bin/bash
# round 4, GPU call 3: more h / G rows in AGPRs for the layouts whose h lives in scratch (NR_HG_REG_ROWS 12 -> 15 / 18):
# NR time and live HBM traffic on the big-batch shapes, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_q3}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-shapes"
for cfg in case141:8192 case322:8192 case322:4096; do
  c=${cfg%%:*}; b=${cfg##*:}
  for v in hg12 hg15 hg18; do
    lib=$R/mapdn_amd/lib_$v.so; [ $v = hg12 ] && lib=$R/mapdn_amd/libmapdn_hip.so
    env MAPDN_LIB_PATH=$lib timeout 300 $B --case $c --envs $b > $OUT/bench_${v}_${c}_b$b.json 2>> $OUT/bench.err
  done
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2), 'traffic MB', round((r.get('traffic') or 0)/1e6,1), 'alg MB', round(r['algorithmic_bytes_per_launch']/1e6,1))"
done | tee $OUT/summary.txt
