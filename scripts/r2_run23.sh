#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r3j}
timeout 1200 python3 -m pytest tests/test_gpu_map.py -m gpu -q -x > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  |Error" gpurun_out/${V}_pytest.log | head -30
