#!/bin/bash
# quick GPU check: given pytest selection, then a short bench line.  Writes gpurun_out/$TAG/.
TAG=${1:-quick}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest "$@" -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
cd /tmp
timeout 300 python $R/bench.py --no-cpu-baseline --no-traffic > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
