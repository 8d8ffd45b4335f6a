#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2p}
timeout 1200 python3 -m pytest tests/test_gpu_solve.py -m gpu -q -x > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
timeout 600 python3 scripts/solve_scale.py 256 1000000 256,5000 g 0,5 2>&1 | grep -E "kind=|lcty solve" > gpurun_out/${V}_scale.log
cat gpurun_out/${V}_scale.log
