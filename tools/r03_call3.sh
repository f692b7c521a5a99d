#!/bin/bash
# round-3 GPU call: full GPU suite, bench (default + inject A/B), kernel stats, coarse stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_c3}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json
MAPDN_INJECT_FULL=1 timeout 200 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_injectfull.json 2>> $OUT/bench.err; cut -c1-330 $OUT/bench_injectfull.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/ks.log 2>&1
db=$(find $OUT/ks -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats.txt > /dev/null && python $R/tools/launch_gaps.py $db > $OUT/gaps.txt
rm -rf $OUT/ks
head -20 $OUT/kernel_stats.txt
cd $R
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case141 --envs 4096 > $OUT/stamps_case141.txt 2>&1
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case322 --envs 4096 > $OUT/stamps_case322.txt 2>&1
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case33 --envs 4096 > $OUT/stamps_case33.txt 2>&1
grep -E "row|update|solve end" $OUT/stamps_case141.txt
