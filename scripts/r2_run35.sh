#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r10}
timeout 900 python3 scripts/solve_scale.py 256 1000000 5000 g 0 > gpurun_out/${V}_scale.log 2>&1
grep "kind=" gpurun_out/${V}_scale.log
timeout 1500 python3 -m pytest tests/test_gpu_solve.py -m gpu -q -x 2>&1 | tail -3
