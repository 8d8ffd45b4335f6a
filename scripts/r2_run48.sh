#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r23}
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_streaming.py tests/test_gpu_counted.py tests/test_gpu_map.py tests/test_gpu_explicit.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E " | tail -5
for M in pinned pageable; do
timeout 600 python3 scripts/ont_stream_probe.py 16384 256 8192 $M > gpurun_out/${V}_ont_$M.json 2> gpurun_out/${V}_ont_$M.err
python3 -c "
import json; d=json.loads(open('gpurun_out/${V}_ont_$M.json').read().strip().splitlines()[-1]); print('$M', 'upload_GBs', d['upload_GBs'], 'upload_s', d['upload_s'], 'raw_GB', d['raw_GB'])"
done
