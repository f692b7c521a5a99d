#!/bin/bash
# Round-6 profile batch (the recipe of rounds 4-5, + the end-to-end loop and the critic head's MFMA counters) for profiles/: per shape a bench line (live, calibrated PMC traffic) + kernel stats; the default and the
# driver-style run; SQ counters and cycle stamps of the NR kernel (headline).  args: TAG part...   parts: main rest full driver sq latency suite e2e headpmc
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
cfg_run() {  # case envs
  local c=$1 b=$2 t=${1}_b${2}
  timeout 600 python $R/bench.py --case $c --envs $b --steps 480 --warmup 24 --no-cpu-baseline --no-other-shapes > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_$t -o ks -- python $R/bench.py --case $c --envs $b --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-other-shapes > /dev/null 2> $OUT/ks_$t.log
  db=$(find $OUT/ks_$t -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_$t.txt > /dev/null
  rm -rf $OUT/ks_$t
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$t.json").read().strip().splitlines()[-1]); r = d["roofline"]; td = r.get("traffic_detail") or {}
print("$t", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"] * 1e3, 1), "us; nr", round(r["kernel_avg_ms"] * 1e3, 1), "us frac", round(r["frac"], 4),
      "traffic MB", round((r["traffic"] or 0) / 1e6, 1), "alg MB", round(r["algorithmic_bytes_per_launch"] / 1e6, 1))
json.dump({"case_envs": "$t", "traffic_bytes_per_launch": td.get("bytes"), "fetch_bytes": td.get("fetch_bytes"), "write_bytes": td.get("write_bytes"),
           "bytes_raw_counters": td.get("bytes_raw"), "launches": td.get("launches"), "source": td.get("source"),
           "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]}, open("$OUT/traffic_$t.json", "w"))
PY
}
for part in "$@"; do
case $part in
main) cfg_run case141 4096; cfg_run case322 1024;;
rest) cfg_run case33 4096; cfg_run case322 8192; cfg_run case141_deep 4096;;
latency) timeout 120 python $R/tools/dropin_latency.py > $OUT/dropin_latency_b1.txt 2>&1; tail -3 $OUT/dropin_latency_b1.txt;;
suite) cd $R; timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.txt 2>&1; tail -3 $OUT/gpu_suite.txt; cd /tmp;;
full) timeout 900 python $R/bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default_run.json;;
driver) timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style_steps20.json 2> $OUT/bench_driver.err; cut -c1-300 $OUT/bench_driver_style_steps20.json
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_drv -o ks -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > /dev/null 2> $OUT/ks_drv.log
    db=$(find $OUT/ks_drv -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/kernel_stats_driver_command.txt > /dev/null; rm -rf $OUT/ks_drv;;
sq) timeout 600 rocprofv3 -i $R/tools/pmc_sq.txt --output-format csv -d $OUT/sq -o sq -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-traffic --no-other-shapes > $OUT/sq.log 2>&1
    python $R/tools/pmc_sq_summary.py --kernel k_nr_tree $OUT/nr_sq_counters.txt $(find $OUT/sq -name "*counter_collection.csv") | head -24; rm -rf $OUT/sq;;
e2e) timeout 600 python $R/examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 4 --intensity reference --phases --log $OUT/e2e_maddpg_case322_b8192_reference.jsonl > $OUT/e2e.log 2>&1; tail -1 $OUT/e2e.log | cut -c1-500
    timeout 600 python $R/examples/train_ddpg.py --case case141 --envs 4096 --alg iddpg --episodes 3 --intensity reference --phases --log $OUT/e2e_iddpg_case141_b4096_reference.jsonl > $OUT/e2e_iddpg.log 2>&1; tail -1 $OUT/e2e_iddpg.log | cut -c1-400
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/ks_e2e -o ks -- python $R/examples/train_ddpg.py --case case322 --envs 8192 --alg maddpg --episodes 2 --intensity reference > /dev/null 2> $OUT/ks_e2e.log
    db=$(find $OUT/ks_e2e -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/prof_summary.py $db $OUT/e2e_reference_kernel_stats.txt > /dev/null; rm -rf $OUT/ks_e2e; head -30 $OUT/e2e_reference_kernel_stats.txt | cut -c1-140;;
headpmc) python $R/tools/head_bench.py 2>/dev/null | grep rows > $OUT/head_bench.txt
    for pm in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
      tag=$(echo $pm | cut -d' ' -f1)
      timeout 300 rocprofv3 --pmc $pm --output-format csv -d $OUT/pm_$tag -o pm -- python $R/tools/head_bench.py > /dev/null 2> $OUT/pm_$tag.log
      python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$OUT/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_head_" in k and "reduce" not in k:
            v[(k.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/head_mfma_counters.txt", "a") as o:
    for (k, c), x in sorted(v.items()):
        line = f"{k:42s} {c:28s} per launch {sum(x)/len(x):18.1f}   launches {len(x)}"
        print(line); o.write(line + "\n")
PY
      rm -rf $OUT/pm_$tag
    done
    cat $OUT/head_bench.txt;;
stamps) for c in case141 case141_deep case322 case33; do MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python $R/tools/nr_stamps.py --case $c --envs 4096 --step > $OUT/stamps_$c.txt 2>&1; done; tail -30 $OUT/stamps_case141.txt;;
esac
done
ls $OUT | head -50
