#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_c8}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json; echo
MAPDN_NR_MM_PASS=0 timeout 200 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_mmsweep.json 2>> $OUT/bench.err; cut -c1-330 $OUT/bench_mmsweep.json; echo
grep -o '"kernel_avg_ms": [0-9.]*' $OUT/bench.json $OUT/bench_mmsweep.json
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case141 --envs 4096 > $OUT/stamps_case141.txt 2>&1
grep -E "row|update|solve end|verdict" $OUT/stamps_case141.txt | tail -12
