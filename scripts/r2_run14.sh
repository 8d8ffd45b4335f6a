#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2n}
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_solve.py -m gpu -q -k "sharded or many_locations or greedy_and_anneal" > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
bash scripts/r2_run13.sh $V
