#!/bin/bash
# round 5, questions 2 + 3: tolerance-edge tests, RCCL first contact at one rank, then the edge soak (>= 10^6 power flows)
mkdir -p gpurun_out; O=gpurun_out/r05_q2.txt; : > $O
nproc | tee -a $O
timeout 900 python -m pytest tests/test_nr_tolerance_edge.py tests/test_bench_selflaunch.py::test_bench_force_dist_rccl_first_contact -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -25 | tee -a $O
timeout 300 python bench.py --force-dist --no-traffic --no-cpu-baseline --no-other-shapes > gpurun_out/r05_bench_force_dist.json 2> gpurun_out/r05_bench_force_dist.err; tail -c 600 gpurun_out/r05_bench_force_dist.json | tee -a $O
S=gpurun_out/r05_edge_soak.txt; : > $S
timeout 1500 python tools/edge_soak.py --case case141 --n 1048576 2>> gpurun_out/r05_edge_soak.err | tee -a $S
timeout 600 python tools/edge_soak.py --case case33 --n 262144 2>> gpurun_out/r05_edge_soak.err | tee -a $S
timeout 900 python tools/edge_soak.py --case case322 --n 131072 2>> gpurun_out/r05_edge_soak.err | tee -a $S
