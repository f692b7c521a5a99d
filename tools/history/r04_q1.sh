#!/bin/bash
# round 4, GPU call 1: full GPU suite on the refactored build, then same-box A/B of the step composition on case141 x 4096:
#   base   = library built WITHOUT the fused prologue (-DMAPDN_NO_FUSE_PROLOGUE), injection as its own launch  (round-3 form)
#   nofuse = product library, fuse_inject off      (register-allocation side effect of the prologue alone)
#   fuse   = product library, default (PV-bus injection in the k_nr_tree prologue)
#   overlap= product library, fuse off, profile rows of k_advance on a side stream beside the solver launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_q1}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-shapes --no-traffic"
run() { tag=$1; shift; env "$@" timeout 200 $B > $OUT/bench_$tag.json 2>> $OUT/bench.err; }
for rep in 1 2; do
  run base_$rep MAPDN_LIB_PATH=$R/mapdn_amd/lib_nofuse.so MAPDN_FUSE_INJECT=0
  run nofuse_$rep MAPDN_FUSE_INJECT=0
  run fuse_$rep MAPDN_X=1
  run overlap_$rep MAPDN_FUSE_INJECT=0 MAPDN_OVERLAP_ADVANCE=1
  run fuse_overlap_$rep MAPDN_OVERLAP_ADVANCE=1
done
for cfg in case322:1024 case322:8192 case33:4096 case141:8192; do
  c=${cfg%%:*}; b=${cfg##*:}
  env MAPDN_FUSE_INJECT=0 timeout 200 $B --case $c --envs $b > $OUT/bench_nofuse_${c}_b$b.json 2>> $OUT/bench.err
  timeout 200 $B --case $c --envs $b > $OUT/bench_fuse_${c}_b$b.json 2>> $OUT/bench.err
done
rocprofv3 --kernel-trace --stats -d $OUT/prof_fuse -o ks -- $B --steps 240 --min-seconds 0.2 > /dev/null 2>> $OUT/bench.err
db=$(find $OUT/prof_fuse -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_fuse.txt > /dev/null; rm -rf $OUT/prof_fuse
timeout 120 python $R/tools/policy_bench.py > $OUT/policy_bench.txt 2>&1
for f in $OUT/bench_*.json; do
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us nr', round(r['kernel_avg_ms']*1e3,2))"
done | tee $OUT/summary.txt
