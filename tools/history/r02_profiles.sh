#!/bin/bash
# Round-2 profile batch for profiles/: per config bench line (live PMC traffic) + kernel stats; SQ counters of the NR kernel
# (headline config); the general-topology kernel with MFMA counters.  args: TAG part...   parts: main dense sq e2e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
cfg_run() {  # case envs
  local c=$1 b=$2 t=${1}_b${2}
  timeout 600 python $R/bench.py --case $c --envs $b --steps 480 --warmup 24 --no-cpu-baseline > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_$t -o ks -- python $R/bench.py --case $c --envs $b --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > /dev/null 2> $OUT/ks_$t.log
  db=$(find $OUT/ks_$t -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_$t.txt > /dev/null
  rm -rf $OUT/ks_$t
  python -c "import json; d=json.load(open('$OUT/bench_$t.json')); r=d['roofline']; print('$t', round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,1),'us; nr', round(r['kernel_avg_ms']*1e3,1),'us frac', round(r['frac'],4), 'traffic', r['traffic'])"
}
for part in "$@"; do
case $part in
main) cfg_run case141 4096; cfg_run case33 4096; cfg_run case322 4096; cfg_run case322 1024; cfg_run case322 8192; cfg_run case141 8192;;
full) timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-300;;
sq) timeout 600 rocprofv3 -i $R/tools/pmc_sq.txt --output-format csv -d $OUT/sq -o sq -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/sq.log 2>&1
    python $R/tools/pmc_sq_summary.py --kernel k_nr_tree $OUT/nr_sq_counters.txt $(find $OUT/sq -name "*counter_collection.csv") | head -24; rm -rf $OUT/sq;;
dense) rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counter_names.txt
    MAPDN_NR_DENSE=1 timeout 300 python $R/tools/general_bench.py > $OUT/dense_bench_case33_meshed5.json 2> $OUT/dense_bench.err; cat $OUT/dense_bench_case33_meshed5.json
    MAPDN_NR_DENSE=1 timeout 300 python $R/tools/general_bench.py --ties 0 > $OUT/dense_bench_case33_radial_forced.json 2>> $OUT/dense_bench.err
    timeout 300 python $R/bench.py --case case33 --envs 4096 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/tree_bench_case33.json 2>> $OUT/dense_bench.err
    MAPDN_NR_DENSE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ksd -o ks -- python $R/tools/general_bench.py > /dev/null 2> $OUT/ksd.log
    db=$(find $OUT/ksd -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_dense_case33_meshed5.txt > /dev/null; rm -rf $OUT/ksd
    for pm in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
      tag=$(echo $pm | cut -d' ' -f1)
      MAPDN_NR_DENSE=1 timeout 300 rocprofv3 --pmc $pm --output-format csv -d $OUT/pmd_$tag -o pm -- python $R/tools/general_bench.py --steps 20 > /dev/null 2> $OUT/pmd_$tag.log
      python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$OUT/pmd_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_nr_dense" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/dense_mfma_counters.txt", "a") as o:
    for k, x in sorted(v.items()):
        x = [y for y in x if y > 0.25 * max(x)] if max(x) > 0 else x
        line = f"{k:34s} per launch {sum(x)/len(x):16.1f}   launches {len(x)}"
        print(line); o.write(line + "\n")
PY
      rm -rf $OUT/pmd_$tag
    done;;
sparse) for cfg in "case33 5" "case141 5" "case322 5"; do set -- $cfg
      timeout 300 python $R/tools/general_bench.py --case $1 --ties $2 > $OUT/sparse_bench_$1_meshed$2.json 2>> $OUT/sparse_bench.err; cat $OUT/sparse_bench_$1_meshed$2.json
      MAPDN_NR_SPARSE=1 timeout 300 python $R/tools/general_bench.py --case $1 --ties 0 > $OUT/sparse_bench_$1_radial_forced.json 2>> $OUT/sparse_bench.err; cat $OUT/sparse_bench_$1_radial_forced.json
      timeout 300 python $R/tools/general_bench.py --case $1 --ties 0 > $OUT/tree_bench_$1.json 2>> $OUT/sparse_bench.err; cat $OUT/tree_bench_$1.json
    done
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kss -o ks -- python $R/tools/general_bench.py --case case141 --ties 5 > /dev/null 2> $OUT/kss.log
    db=$(find $OUT/kss -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_sparse_case141_meshed5.txt > /dev/null; rm -rf $OUT/kss;;
e2e) timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kse -o ks -- python $R/examples/train_ddpg.py --case case322 --envs 8192 --episodes 1 --max-steps 120 > $OUT/e2e.log 2>&1
    db=$(find $OUT/kse -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/e2e_kernel_stats.txt | head -30; rm -rf $OUT/kse; tail -5 $OUT/e2e.log;;
esac
done
ls $OUT
