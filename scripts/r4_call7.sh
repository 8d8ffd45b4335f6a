#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4_c7
timeout 600 python3 scripts/solve_probe.py default solve_greedy_form=32 > gpurun_out/r4_c7/probe.log 2>&1
echo "probe rc=$?"; grep -v '^\[lcty solve' gpurun_out/r4_c7/probe.log | tail -5
timeout 1500 python3 -m pytest tests/test_gpu_solve.py tests/test_gpu_exact.py tests/test_gpu_comm_failures.py tests/test_gpu_bench_launch.py tests/test_gpu_counted.py -m gpu -x -q > gpurun_out/r4_c7/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r4_c7/pytest.log
