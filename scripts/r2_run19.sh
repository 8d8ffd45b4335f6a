#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2x}
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -q -x -k "matrix_cores or page_locked or prefilter" > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -30
timeout 900 python3 scripts/gram_probe.py 32768 1024 > gpurun_out/${V}_gram_1024.json 2> gpurun_out/${V}_gram.err; cat gpurun_out/${V}_gram_1024.json; tail -3 gpurun_out/${V}_gram.err
