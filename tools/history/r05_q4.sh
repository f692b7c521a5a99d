#!/bin/bash
# NOTE: kept as the record of how the experiment was run — the debug library / switch it uses (lib_ab32.so, MAPDN_NR_PAIRS, lib_defer.so) was removed
# again once the result was in profiles/ (f32 mirror: r05_f32_obs_mirror_ab.txt; chain pairs: commit 2ce82a4 + r05_chain_pair_fusion_experiment.txt;
# deferred update: r05_xprop_deferred_update_ab.txt).  It does not run against the current tree.
# round 5, question 4: what would an f32 obs mirror buy?  Timing A/B with the debug library mapdn_amd/lib_ab32.so (built with
# MAPDN_EXTRA_FLAGS="-DMAPDN_DEBUG_BUILD -DMAPDN_AB_F32MIRROR"): the obs gather reading 4-byte columns, the commit rows writing the
# mirror.  Same box, same process order, alternating.
mkdir -p gpurun_out/r05_q4; O=gpurun_out/r05_q4.txt; : > $O
B="python bench.py --no-cpu-baseline --no-traffic --no-other-shapes --steps 240 --min-seconds 0.4"
one() { tag=$1; shift; env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', 'value %.3f M  ms_per_step %.5f  nr %.2f us' % (j['value']/1e6, j['ms_per_step'], j['roofline']['kernel_avg_ms']*1e3))" | tee -a $O; }
L=$PWD/mapdn_amd/lib_ab32.so
for rep in 1 2 3; do
  one base_product_lib MAPDN_X=1
  one ab_lib_off MAPDN_LIB_PATH=$L
  one gather32 MAPDN_LIB_PATH=$L MAPDN_AB_GATHER32=1
  one gather32+mirror_write MAPDN_LIB_PATH=$L MAPDN_AB_GATHER32=1 MAPDN_AB_MIRROR=1
done
cd /tmp && export TMPDIR=/tmp
for v in "off MAPDN_X=1" "both MAPDN_AB_GATHER32=1 MAPDN_AB_MIRROR=1"; do set -- $v; tag=$1; shift
  env MAPDN_LIB_PATH=$L "$@" rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_q4/prof_$tag -o ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-traffic --no-other-shapes --steps 240 --min-seconds 0.2 > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/r05_q4/prof_$tag/*kernel_stats.csv 2>/dev/null | head -1); echo "== kernel stats $tag" | tee -a $GRAFT_REPO_ROOT/$O; head -8 $f | cut -c1-160 | tee -a $GRAFT_REPO_ROOT/$O
done
