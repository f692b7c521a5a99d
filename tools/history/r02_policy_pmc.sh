#!/bin/bash
# MFMA / VALU counters of the fused policy forward (tools/policy_bench.py).  Writes gpurun_out/$TAG/policy_mfma_counters.txt
TAG=${1:-polpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $R/tools/policy_bench.py 2>/dev/null | grep agents > $OUT/policy_bench.txt
for pm in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pm | cut -d' ' -f1)
  MAPDN_FUSED_POLICY=1 timeout 300 rocprofv3 --pmc $pm --output-format csv -d $OUT/pm_$tag -o pm -- python $R/tools/policy_bench.py > /dev/null 2> $OUT/pm_$tag.log
  python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$OUT/pm_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_policy_fwd" in r["Kernel_Name"]:
            v[(r["Counter_Name"], r["Grid_Size"] if "Grid_Size" in r else "")].append(float(r["Counter_Value"]))
with open("$OUT/policy_mfma_counters.txt", "a") as o:
    for (k, g), x in sorted(v.items()):
        line = f"{k:30s} grid {g:>8s}  per launch {sum(x)/len(x):16.1f}   launches {len(x)}"
        print(line); o.write(line + "\n")
PY
  rm -rf $OUT/pm_$tag
done
cat $OUT/policy_bench.txt
