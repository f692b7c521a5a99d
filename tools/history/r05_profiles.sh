#!/bin/bash
# Round-5 profile batch (same recipe as round 4) for profiles/: per shape a bench line (live, calibrated PMC traffic) + kernel stats; the default and the
# driver-style run; SQ counters and cycle stamps of the NR kernel (headline).  args: TAG part...   parts: main full driver sq stamps
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
stamps) for c in case141 case141_deep case322 case33; do MAPDN_LIB_PATH=$R/mapdn_amd/lib_stamps.so timeout 120 python $R/tools/nr_stamps.py --case $c --envs 4096 --step > $OUT/stamps_$c.txt 2>&1; done; tail -30 $OUT/stamps_case141.txt;;
esac
done
ls $OUT | head -50
