#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r05_q9.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bus_fusion.py tests/test_replay.py tests/test_learner.py -x -q -m gpu -k "episode_parity or step_composition or fused_net or replay or learner or window" 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O
MAPDN_NR_WAVES=4 MAPDN_NR_LANES=4 timeout 600 python tools/parity_soak.py --case case141 --envs 1024 --watch 512 --steps 36 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900 | tee -a $O
MAPDN_NR_WAVES=2 MAPDN_NR_LANES=4 timeout 600 python tools/parity_soak.py --case case322 --envs 512 --watch 256 --steps 24 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900 | tee -a $O
timeout 300 python examples/train_ddpg.py --case case322 --envs 8192 --episodes 3 --phases 2>/dev/null | tail -1 | cut -c1-900 | tee -a $O
