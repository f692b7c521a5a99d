#!/bin/bash
# round 6: tables at the real data's scale on the GPU + the reference-class long-table pin through the HIP drop-in
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_real_scale_tables.py "tests/test_env_reference_pin.py" -x -q -m gpu -k "three_year or long" --durations=5 > gpurun_out/r06_q4_real_scale.txt 2>&1; echo "rc=$?"; tail -15 gpurun_out/r06_q4_real_scale.txt
cat gpurun_out/r06_real_scale_*.json
free -g | head -2
