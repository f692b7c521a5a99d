#!/bin/bash
# round 4: XCD-aligned env order of the wide kernels (xcd_map) — parity subset, then same-box A/B per shape (MAPDN_XCD_MAP=0 vs default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_q6}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shipped_configs.py tests/test_env_reference_pin.py tests/test_bus_fusion.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic"
for rep in 1 2; do
  env MAPDN_XCD_MAP=0 timeout 200 $B > $OUT/bench_plain_$rep.json 2>> $OUT/bench.err
  timeout 200 $B > $OUT/bench_xcd_$rep.json 2>> $OUT/bench.err
done
for cfg in case322:1024 case322:8192 case33:4096 case141:8192 case141_deep:4096; do
  c=${cfg%%:*}; b=${cfg##*:}
  env MAPDN_XCD_MAP=0 timeout 200 $B --case $c --envs $b > $OUT/bench_plain_${c}_b$b.json 2>> $OUT/bench.err
  timeout 200 $B --case $c --envs $b > $OUT/bench_xcd_${c}_b$b.json 2>> $OUT/bench.err
done
for v in plain xcd; do
  x=1; [ $v = plain ] && x=0
  env MAPDN_XCD_MAP=$x rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o ks -- $B --steps 240 --min-seconds 0.2 > /dev/null 2>> $OUT/bench.err
  db=$(find $OUT/prof_$v -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_$v.txt > /dev/null; rm -rf $OUT/prof_$v
  head -5 $OUT/kernel_stats_$v.txt | cut -c1-130
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done | tee $OUT/summary.txt
