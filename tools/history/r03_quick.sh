#!/bin/bash
# quick A/B on the GPU box: parity subset, then the case141 x 4096 bench line with live HBM traffic (+ case322 / case33 lines)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_q}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shipped_configs.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
timeout 300 python $R/bench.py --no-cpu-baseline --no-other-shapes > $OUT/bench_default.json 2>> $OUT/bench.err
for cfg in case322:4096 case33:4096 case141:8192; do
  c=${cfg%%:*}; b=${cfg##*:}
  timeout 200 python $R/bench.py --case $c --envs $b --no-cpu-baseline --no-other-shapes > $OUT/bench_${c}_b$b.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us nr', round(r['kernel_avg_ms']*1e3,1), 'traffic', r.get('traffic'), 'alg', r.get('algorithmic_bytes_per_launch'), r.get('traffic_source'))"
done
