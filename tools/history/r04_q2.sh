#!/bin/bash
# round 4, GPU call 2: fused-prologue v2 (batched loads) — parity of the fused injection, then same-box A/B per shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_q2}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_injection or two_handles or every_nr_launch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic"
for rep in 1 2; do
  env MAPDN_LIB_PATH=$R/mapdn_amd/lib_nofuse.so MAPDN_FUSE_INJECT=0 timeout 200 $B > $OUT/bench_base_$rep.json 2>> $OUT/bench.err
  timeout 200 $B > $OUT/bench_fuse_$rep.json 2>> $OUT/bench.err
done
for cfg in case322:1024 case322:8192 case33:4096 case141:8192 case141_deep:4096; do
  c=${cfg%%:*}; b=${cfg##*:}
  env MAPDN_FUSE_INJECT=0 timeout 200 $B --case $c --envs $b > $OUT/bench_nofuse_${c}_b$b.json 2>> $OUT/bench.err
  timeout 200 $B --case $c --envs $b > $OUT/bench_fuse_${c}_b$b.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done | tee $OUT/summary.txt
