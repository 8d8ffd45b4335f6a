#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r2k}
timeout 900 python3 -m pytest tests/test_gpu_bam.py -q -x > gpurun_out/${V}_pytest_bam.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest_bam.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest_bam.log | head -20
