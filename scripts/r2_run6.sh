#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2f}
timeout 900 python3 scripts/solve_scale.py 256 1000000 256,1024,2048,4096,5000 g 4,2,1 > gpurun_out/${V}_scale.log 2>&1
cat gpurun_out/${V}_scale.log | tail -20
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
