#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2g}
timeout 900 python3 -m pytest tests/test_gpu_solve.py tests/test_gpu_counted.py tests/test_gpu_pins.py tests/test_gpu_debug_dumps.py -q > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
timeout 900 python3 scripts/solve_scale.py 256 1000000 256,5000 g 4 > gpurun_out/${V}_scale.log 2>&1
tail -3 gpurun_out/${V}_scale.log
