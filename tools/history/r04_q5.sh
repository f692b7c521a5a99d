#!/bin/bash
# round 4: TIMING experiment — k_advance and the obs gather of one step as ONE launch (results wrong: no commit -> gather order),
# upper bound on what overlapping the two wide kernels can give; plus a parity subset on the refactored product build
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_q5}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_env_reference_pin.py tests/test_reference_loop.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic"
for rep in 1 2; do
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_merged.so timeout 200 $B > $OUT/bench_separate_$rep.json 2>> $OUT/bench.err
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_merged.so MAPDN_EXP_MERGED=1 timeout 200 $B > $OUT/bench_merged_$rep.json 2>> $OUT/bench.err
done
for cfg in case322:8192 case33:4096; do
  c=${cfg%%:*}; b=${cfg##*:}
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_merged.so timeout 200 $B --case $c --envs $b > $OUT/bench_separate_${c}_b$b.json 2>> $OUT/bench.err
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_merged.so MAPDN_EXP_MERGED=1 timeout 200 $B --case $c --envs $b > $OUT/bench_merged_${c}_b$b.json 2>> $OUT/bench.err
done
env MAPDN_LIB_PATH=$R/mapdn_amd/lib_merged.so MAPDN_EXP_MERGED=1 rocprofv3 --kernel-trace --stats -d $OUT/prof -o ks -- $B --steps 240 --min-seconds 0.2 > /dev/null 2>> $OUT/bench.err
db=$(find $OUT/prof -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_merged.txt > /dev/null; rm -rf $OUT/prof
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done | tee $OUT/summary.txt
head -7 $OUT/kernel_stats_merged.txt
