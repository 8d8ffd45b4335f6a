#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2q}
timeout 600 python3 scripts/solve_scale.py 256 1000000 5000 g 4,2,1 2>&1 | grep -E "kind=" > gpurun_out/${V}_scale.log
cat gpurun_out/${V}_scale.log
