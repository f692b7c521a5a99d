#!/bin/bash
# round 4: k_nr_dense with the Jacobian in global memory (141- / 322-bus nets): parity tests, throughput, MFMA counters (X1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r04_dense}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_general_topology.py tests/test_literature_pins.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
cd /tmp
for cfg in "case141 5 1024 6" "case141 0 1024 6" "case322 5 256 4"; do set -- $cfg
  MAPDN_NR_DENSE=1 timeout 400 python $R/tools/general_bench.py --case $1 --ties $2 --envs $3 --steps $4 > $OUT/dense_bench_$1_ties$2_b$3.json 2>> $OUT/dense_bench.err; cat $OUT/dense_bench_$1_ties$2_b$3.json
done
MAPDN_NR_DENSE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/ksd -o ks -- python $R/tools/general_bench.py --case case141 --ties 5 --envs 1024 --steps 4 > /dev/null 2> $OUT/ksd.log
db=$(find $OUT/ksd -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_dense_case141_meshed5_b1024.txt > /dev/null; rm -rf $OUT/ksd
head -5 $OUT/kernel_stats_dense_case141_meshed5_b1024.txt | cut -c1-140
: > $OUT/dense_case141_mfma_counters.txt
for pm in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pm | cut -d' ' -f1)
  MAPDN_NR_DENSE=1 timeout 400 rocprofv3 --pmc $pm --output-format csv -d $OUT/pmd_$tag -o pm -- python $R/tools/general_bench.py --case case141 --ties 5 --envs 1024 --steps 3 > /dev/null 2> $OUT/pmd_$tag.log
  python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$OUT/pmd_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_nr_dense" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/dense_case141_mfma_counters.txt", "a") as o:
    for k, x in sorted(v.items()):
        x = [y for y in x if y > 0.25 * max(x)] if max(x) > 0 else x
        line = f"{k:34s} per launch {sum(x)/len(x):18.1f}   launches {len(x)}"
        print(line); o.write(line + "\n")
PY
  rm -rf $OUT/pmd_$tag
done
