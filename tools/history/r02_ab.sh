#!/bin/bash
# A/B on the GPU box: selected GPU tests, then kernel trace of a short bench run (per-kernel avg + gaps) for the given cases.
TAG=${1:-ab}; SEL=${2:-tests/test_gpu_parity.py}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest $SEL -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
cd /tmp
for cfg in "$@"; do
  c=${cfg%%:*}; b=${cfg##*:}
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_$c$b -o ks -- python $R/bench.py --case $c --envs $b --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/bench_$c$b.json 2> $OUT/ks_$c$b.log
  db=$(find $OUT/ks_$c$b -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/launch_gaps.py $db > $OUT/gaps_$c$b.txt && python $R/tools/prof_summary.py $db $OUT/kernel_stats_$c$b.txt > /dev/null
  rm -rf $OUT/ks_$c$b
  echo "== $c B=$b"; python -c "import json; d=json.loads(open('$OUT/bench_$c$b.json').read().strip().splitlines()[-1]); print('bench(profiled)', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step']*1e3,1), 'us')"
  head -6 $OUT/gaps_$c$b.txt
done
