#!/bin/bash
# Round-2 profile sweep (VERDICT item 6 ii): kernel stats + FETCH/WRITE PMC passes of bench.py for the
# configurations that had none in round 1.  Run on the GPU box from the repo root; writes gpurun_out/$TAG/.
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # case envs
  local c=$1 b=$2 t=${1}_b${2}
  python $R/bench.py --case $c --envs $b --steps 240 --warmup 12 --no-cpu-baseline > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  rocprofv3 --kernel-trace --stats -d $OUT/ks_$t -o ks -- python $R/bench.py --case $c --envs $b --steps 100 --warmup 10 --no-cpu-baseline > $OUT/ks_$t.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_$t -o f -- python $R/bench.py --case $c --envs $b --steps 60 --warmup 5 --no-cpu-baseline > $OUT/f_$t.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w_$t -o w -- python $R/bench.py --case $c --envs $b --steps 60 --warmup 5 --no-cpu-baseline > $OUT/w_$t.log 2>&1
  python $R/tools/pmc_traffic.py $(dirname $(find $OUT/f_$t -name "*counter_collection.csv" | head -1)) $(dirname $(find $OUT/w_$t -name "*counter_collection.csv" | head -1)) $OUT/traffic_$t.json > /dev/null 2>> $OUT/bench_$t.err
  db=$(find $OUT/ks_$t -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_$t.txt > /dev/null
  # keep only the small summaries (the merge-back limit is 64 MiB)
  rm -rf $OUT/ks_$t $OUT/f_$t $OUT/w_$t
}
shift
for cfg in "$@"; do run ${cfg%%:*} ${cfg##*:}; done
ls -la $OUT
