#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
V=${1:-r4k}
timeout 1200 python3 -m pytest tests/test_gpu_map.py -m gpu -q -x > gpurun_out/${V}_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/${V}_pytest.log | tail -2; grep -E "^FAILED|^ERROR|^E  " gpurun_out/${V}_pytest.log | head -20
timeout 900 python3 bench.py --steps 1 --warmup 0 --pairs 65536 --cpu-sample 0 --recovery-sample 0 --recruit-sample 0 --many-alleles-sample 0 --ont-sample 0 > gpurun_out/${V}_map.json 2> gpurun_out/${V}_map.err
python3 -c "
import json; d=json.load(open('gpurun_out/${V}_map.json')); print(d.get('candidate_generation'))"
