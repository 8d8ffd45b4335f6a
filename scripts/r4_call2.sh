#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4_c2
timeout 600 python3 scripts/solve_probe.py default solve_greedy_form=1 solve_greedy_form=2 solve_greedy_form=4 solve_greedy_form=8 solve_greedy_form=16 solve_greedy_form=24 solve_greedy_form=31 default > gpurun_out/r4_c2/probe.log 2>&1
echo "probe rc=$?"; grep -v '^\[' gpurun_out/r4_c2/probe.log | tail -12
timeout 1200 python3 -m pytest tests/test_gpu_solve.py tests/test_gpu_exact.py tests/test_gpu_comm_failures.py tests/test_gpu_bench_launch.py -m gpu -x -q > gpurun_out/r4_c2/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r4_c2/pytest.log
