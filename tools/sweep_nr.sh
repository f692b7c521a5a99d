#!/bin/bash
# sweep NR launch geometry (waves per env group W, envs per wave L) for one case / batch size
CASE=${1:-case141}; ENVS=${2:-4096}; STEPS=${3:-120}
for W in 1 2 4 8 16; do for L in 64 32 16; do
  if [ "$W" = "0" ] && [ "$L" != "64" ]; then continue; fi
  MAPDN_NR_WAVES=$W MAPDN_NR_LANES=$L python bench.py --case $CASE --envs $ENVS --steps $STEPS --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CASE B=$ENVS W=$W L=$L  steps/s=%.3e  ms/step=%.3f  nr_ms=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done; done
