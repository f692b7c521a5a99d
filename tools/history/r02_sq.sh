#!/bin/bash
# SQ counter passes (tools/pmc_sq.txt) of a short bench run, summarised per kernel.  args: TAG case envs kernel-substrings...
TAG=$1; CASE=$2; ENVS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 -i $R/tools/pmc_sq.txt --output-format csv -d $OUT/sq -o sq -- python $R/bench.py --case $CASE --envs $ENVS --steps 30 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/sq.log 2>&1
files=$(find $OUT/sq -name "*counter_collection.csv")
for k in "$@"; do python $R/tools/pmc_sq_summary.py --kernel $k $OUT/sq_$k.txt $files > /dev/null; done
rm -rf $OUT/sq
head -30 $OUT/sq_*.txt
