# same-box A/B of an environment-variable switch of the product library, interleaved repetitions + one kernel table per setting:
#   bash tools/r04_ab_env.sh <out-name> <VAR> "<value> ..." "<case>:<envs> ..." [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; VAR=$2; VALS=$3; SHAPES=${4:-case141:4096}; REPS=${5:-3}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for rep in $(seq 1 $REPS); do
  for sh in $SHAPES; do
    c=${sh%%:*}; b=${sh##*:}
    for v in $VALS; do
      env $VAR=$v timeout 200 python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic --case $c --envs $b > $OUT/bench_${VAR}${v}_${c}_b${b}_$rep.json 2>> $OUT/bench.err
    done
  done
done
for sh in $SHAPES; do
  c=${sh%%:*}; b=${sh##*:}
  for v in $VALS; do
    env $VAR=$v rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o ks -- python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic --case $c --envs $b --steps 240 --min-seconds 0.2 > /dev/null 2>> $OUT/bench.err
    db=$(find $OUT/prof_$v -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_${VAR}${v}_${c}_b${b}.txt > /dev/null; rm -rf $OUT/prof_$v
    echo "== $VAR=$v $c x $b"; head -6 $OUT/kernel_stats_${VAR}${v}_${c}_b${b}.txt | tail -4 | cut -c1-130
  done
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done
