#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r4b}
timeout 1200 python3 -m pytest tests/test_gpu_solve.py -m gpu -q -x -k "greedy_and_anneal or batches or scheme or assignment" > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
timeout 600 python3 scripts/solve_scale.py 256 1000000 5000 g 0 2>&1 | grep -E "kind=" 
