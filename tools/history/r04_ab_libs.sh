# same-box A/B of library variants (mapdn_amd/lib_<name>.so; "head" = the product library), interleaved repetitions:
#   bash tools/r04_ab_libs.sh <out-name> "<variant> ..." "<case>:<envs> ..." [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; VARS=$2; SHAPES=${3:-case141:4096}; REPS=${4:-3}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for rep in $(seq 1 $REPS); do
  for sh in $SHAPES; do
    c=${sh%%:*}; b=${sh##*:}
    for v in $VARS; do
      lib=$R/mapdn_amd/lib_$v.so; [ $v = head ] && lib=$R/mapdn_amd/libmapdn_hip.so
      env MAPDN_LIB_PATH=$lib timeout 200 python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic --case $c --envs $b > $OUT/bench_${v}_${c}_b${b}_$rep.json 2>> $OUT/bench.err
    done
  done
done
sh=${SHAPES%% *}; c=${sh%%:*}; b=${sh##*:}
for v in $VARS; do                                  # kernel table of the first shape per variant
  lib=$R/mapdn_amd/lib_$v.so; [ $v = head ] && lib=$R/mapdn_amd/libmapdn_hip.so
  env MAPDN_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o ks -- python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic --case $c --envs $b --steps 240 --min-seconds 0.2 > /dev/null 2>> $OUT/bench.err
  db=$(find $OUT/prof_$v -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_$v.txt > /dev/null; rm -rf $OUT/prof_$v
  echo "== $v"; head -6 $OUT/kernel_stats_$v.txt | tail -4 | cut -c1-130
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done
