#!/bin/bash
# round 6: k_nr_tree launched one ROUND of workgroups at a time (wg0 offset) on batches beyond one round; headline unchanged?
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_shipped_configs.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --case $1 --envs $2 --steps 240 --warmup 24 --no-cpu-baseline --no-other-shapes --no-traffic 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['config']['envs_per_gpu'], round(j['value']/1e6,2), 'M', round(j['ms_per_step']*1e3,1), 'us/step; nr', round(j['roofline']['kernel_avg_ms']*1e3,1))
"; }
for C in 1 0; do echo "== MAPDN_NR_CHUNK=$C"; export MAPDN_NR_CHUNK=$C; run case322 8192; run case141 8192; run case141 4096; run case33 16384; done
