#!/bin/bash
# round-3 GPU call 1: full GPU suite (new shipped-config / literature / reference-fixture tests), baseline bench, cycle stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r03_c2}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case141 --envs 4096 --rows > $OUT/stamps_case141.txt 2>&1
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps_fine.so timeout 120 python tools/nr_stamps.py --case case141 --envs 4096 > $OUT/stamps_fine_case141.txt 2>&1
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case322 --envs 4096 > $OUT/stamps_case322.txt 2>&1
MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python tools/nr_stamps.py --case case33 --envs 4096 > $OUT/stamps_case33.txt 2>&1
tail -30 $OUT/stamps_case141.txt; cat $OUT/stamps_fine_case141.txt | tail -25
